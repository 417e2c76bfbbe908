// C-ABI layer (include/rt_amd.h): context, scene validation + flattening into the kernels' record layout,
// launches, staging of host outputs.  No oracle, no CPU rendering fallback: without a gfx950 device every entry
// point that needs one fails with RT_ERR_NO_DEVICE / RT_ERR_HIP.
#include "rt_amd.h"
#include "rt_launch.h"
#include "../host/camera.h"

#include <rccl/rccl.h>      // types and prototypes only: the library is bound with dlopen (see rccl() below)
#include <dlfcn.h>

#include <algorithm>
#include <cmath>
#include <cstring>
#include <ctime>
#include <mutex>
#include <new>
#include <vector>

// sample-parallel pixels: table capacity (pixels, bytes) and the hand-over threshold in
// draws(sample 0) * (samples - 1); a pixel below it is quicker left to its own lane
#ifndef RT_HEAVY_CAP
#define RT_HEAVY_CAP 262144
#endif
// scenes with at least this many hittables render their bounce pixels one wave per pixel (coop_kernel)
#ifndef RT_COOP_MIN_HIT
#define RT_COOP_MIN_HIT 256
#endif
// direct pixels x samples from which a frame on its own is taken to be bound by the direct kernel (see rng_scratch)
#ifndef RT_DIRECT_BOUND_WORK
#define RT_DIRECT_BOUND_WORK 2.5e8
#endif
#ifndef RT_HEAVY_TABLE_BYTES
#define RT_HEAVY_TABLE_BYTES ((size_t)512 << 20)
#endif
#ifndef RT_HEAVY_MIN_WORK
#define RT_HEAVY_MIN_WORK 12000u
#endif
// the same bar under RT_FLAG_SEQUENCE; the environment variable of this name overrides it. Measured (DESIGN.md section 5,
// 3c): 200000 makes glass 1080p 8 % faster with 16 frames in flight and the last frames of a short 4K sequence slower
// (they run alone, where the hand-over is what shortens them), so it stays equal to RT_HEAVY_MIN_WORK.  Round 3: under
// RT_FLAG_SEQUENCE only frames bound by their direct kernel hand pixels over at all (render_rows_impl, no_handover).
#ifndef RT_HEAVY_MIN_WORK_SEQ
#define RT_HEAVY_MIN_WORK_SEQ RT_HEAVY_MIN_WORK
#endif
// frames in flight: the listed kernel runs with a wave per workgroup (LaunchTraits::wave_groups) when the context's last
// frame of this scene kept it busy for at least this long -- then the frames' listed waves are what the SIMDs' slots
// are short of (glass 1080p: 160 - 360 ms per launch, 27.5 -> 22.3 ms per frame), while short listed kernels (the 4K
// frame: 5 ms) lose by it (3.79 -> 3.85 ms).  The environment variable of this name overrides it (0: always).
#ifndef RT_WAVE_GROUPS_MS
#define RT_WAVE_GROUPS_MS 20.0
#endif
struct rt_ctx {
    int device = -1;
    hipStream_t stream = nullptr;      // everything but the listed general kernel
    hipStream_t stream2 = nullptr;     // the listed general kernel, beside the direct kernel
    hipStream_t stream3 = nullptr;     // its second launch (RT_MODE_REST) when long pixels are handed over
    hipEvent_t ev_rest = nullptr;
    hipStream_t shared_used = nullptr;      // the shared stream this context's last direct kernel went to (drain_streams)
    bool counted = false;                   // among g_direct[device].users
    hipEvent_t ev_wait = nullptr;           // wait_stream()
    unsigned long long* h_back = nullptr;   // 8 pinned words: a frame's read-backs (a copy to pageable memory waits inside hipMemcpyAsync, polling)
    hipEvent_t ev0 = nullptr, ev1 = nullptr, evm = nullptr, ev_list = nullptr, ev_gen = nullptr;
    hipEvent_t ev_l0 = nullptr, ev_l1 = nullptr;   // around the listed kernel of a frame in flight
    bool listed_timed = false;                     // ev_l0 / ev_l1 were recorded by the last frame
    float listed_ms = 0.0f;                        // what they measured last (0: nothing yet for this scene)
    float last_ms = 0.0f, last_total_ms = 0.0f, last_gather_ms = 0.0f;
    hipEvent_t ev_g0 = nullptr, ev_g1 = nullptr;   // rt_gather_rows: around the RCCL group and the reassembly
    int last_hip = 0;
    int last_rccl = 0;

    // uploaded scene
    bool has_scene = false;
    int n_hit = 0, n_lights = 0, n_mat = 0;
    bool mixed = false, nonsphere = false;
    void* d_scene = nullptr;       // hrec | lights | mats
    size_t scene_bytes = 0;
    const float4* d_hrec = nullptr; const float4* d_lights = nullptr; const float* d_mats = nullptr;
    float4* d_prim = nullptr;                    // camera-relative sphere terms, refreshed when the camera moves
    const unsigned int* d_order = nullptr;       // hittable indices sorted by type (cooperative scans)
    const unsigned int* d_slot_of = nullptr;     // the inverse: place of every hittable in that order
    const float* d_clusters = nullptr; int n_clusters = 0;     // boxes of the rounds of 64 (centre, half extents, flag)
    std::vector<float> h_centres_r2;             // host copy of (centre, r^2) per hittable
    std::vector<float> h_prim;                   // staging for d_prim, sized by rt_upload_scene (no allocation per frame)
    float prim_cam[3] = { 0, 0, 0 }; bool prim_valid = false;
    unsigned char* scratch = nullptr;            // this GPU's copy of the static scratch array (rt_render_rows_static)

    // small device words: 4 counters (u64) + fault flag
    unsigned long long* d_counters = nullptr;
    int* d_fault = nullptr;

    // staging for host-pointer outputs (grown on demand)
    void* d_stage_rgb = nullptr; size_t stage_rgb_bytes = 0;
    void* d_stage_em = nullptr;  size_t stage_em_bytes = 0;

    // pixel classes: keys (u16 per pixel) | pixel list (u32 per pixel) | histogram + cursor (2 * RT_MAX_KEYS u32)
    void* d_classes = nullptr; size_t classes_bytes = 0;
    void* d_tiles = nullptr; size_t tiles_bytes = 0;   // rt_render_tiles: chunk rows + tile records
    void* d_gather = nullptr; size_t gather_bytes = 0; // rt_render_gather: band block | gather buffer | frame
    void* d_glow = nullptr; size_t glow_bytes = 0;     // rt_apply_glow: level buffers | weights | staging
    std::vector<float> glow_kernels[8];                // ... the blur's weights per pyramid level (host)
    void* d_heavy = nullptr; size_t heavy_bytes = 0;   // sample-parallel pixels: count | records | generator states
    size_t heavy_pool_bytes = 0;                       // ... its stream pool (sized from the pixel list, rt_render_rows)
    bool any_bounce = false;             // some material has reflection > 0 or refraction > 0
};

// device words: 5 tallies (u64) followed by the fault flag
#define RT_DEV_WORDS_BYTES (5 * sizeof(unsigned long long) + sizeof(int) * 2)
// Everything a frame needs zeroed sits in one block behind them, so that one fill per frame does it (under load every
// small dispatch waits its turn for tens of microseconds): the sample-parallel route's counters (heavy_count,
// fallback_count, pool words used: 16 bytes), then the pixel classes' histogram and cursor (2 * RT_MAX_KEYS words).
#define RT_DEV_HEAVY_AT  RT_DEV_WORDS_BYTES
#define RT_DEV_HIST_AT   (RT_DEV_HEAVY_AT + 16)
#define RT_DEV_STATE_BYTES (RT_DEV_HIST_AT + 2 * RT_MAX_KEYS * sizeof(unsigned int))

// rt_render_rows_static (include/rt_amd.h): the per-frame device state and the pixel classes of a call that may not
// allocate live in this array instead of in buffers the context allocates.  One copy per GPU the library runs on, held
// until the process ends.  Layout: [0, RT_STATIC_STATE_BYTES) the RT_DEV_STATE_BYTES block | keys (u16 per pixel) |
// pixel list (u32 per pixel), for at most RT_STATIC_MAX_PIXELS pixels.
#define RT_STATIC_STATE_BYTES 4096
#define RT_STATIC_KEYS_BYTES  ((size_t)RT_STATIC_MAX_PIXELS * 2)
#define RT_STATIC_LIST_BYTES  ((size_t)RT_STATIC_MAX_PIXELS * 4)
#define RT_STATIC_SCRATCH_BYTES (RT_STATIC_STATE_BYTES + RT_STATIC_KEYS_BYTES + RT_STATIC_LIST_BYTES)
static_assert(RT_DEV_STATE_BYTES <= RT_STATIC_STATE_BYTES, "the per-frame state block must fit its slot of the scratch array");
static __device__ __align__(256) unsigned char scratch[RT_STATIC_SCRATCH_BYTES];

// `scratch` above is a host-side stand-in, not a device address: every GPU holds its own copy at its own address, looked
// up once per device (rt_create, with that device current) and kept here.  Host code only ever uses these pointers.
#define RT_MAX_DEVICES 64
static std::mutex g_scratch_mutex;
static unsigned char* g_scratch_ptr[RT_MAX_DEVICES];

// Frames in flight (RT_FLAG_SEQUENCE): the direct kernels of all contexts of a device go round a few streams that the
// contexts share, instead of each waiting on its context's stream behind that frame's listed kernels.  Contexts that
// start together (the first frames of a sequence, the last ones) then have their direct kernels running while their
// listed kernels -- few waves, long chains -- are under way, and the number of busy streams stays near the number of
// hardware queues (16 contexts + 6 on GPU_MAX_HW_QUEUES=16; a second stream per context measured worse, DESIGN.md
// section 6).  0 in the environment variable of the same name: off.
#ifndef RT_SEQ_DIRECT_STREAMS
#define RT_SEQ_DIRECT_STREAMS 6
#endif
#define RT_SEQ_DIRECT_MAX 8
struct SharedDirect { hipStream_t s[RT_SEQ_DIRECT_MAX] = {}; unsigned turn = 0; int users = 0; };
static std::mutex g_direct_mutex;
static SharedDirect g_direct[RT_MAX_DEVICES];

// A failing call leaves the entry point at once; work already queued on the context's streams (kernels, copies into
// the caller's frame or into locals of the returning function) must not outlive it.
static void drain_streams(rt_ctx* ctx)
{
    if (ctx->stream)  (void)hipStreamSynchronize(ctx->stream);
    if (ctx->stream2) (void)hipStreamSynchronize(ctx->stream2);
    if (ctx->stream3) (void)hipStreamSynchronize(ctx->stream3);
    if (ctx->shared_used) (void)hipStreamSynchronize(ctx->shared_used);
}

// streams a context with RT_FLAG_SEQUENCE works on: 1 (listed kernels in front of the direct kernel) or 2 (beside it)
#ifndef RT_SEQ_STREAMS_DEFAULT
#define RT_SEQ_STREAMS_DEFAULT 1
#endif

#define HIP_TRY(ctx, expr)                                   \
    do {                                                     \
        hipError_t e__ = (expr);                             \
        if (e__ != hipSuccess) { (ctx)->last_hip = (int)e__; drain_streams(ctx); return RT_ERR_HIP; } \
    } while (0)

namespace {

struct DeviceGuard {
    int prev = -1; bool ok = false;
    explicit DeviceGuard(int dev) { ok = hipGetDevice(&prev) == hipSuccess && hipSetDevice(dev) == hipSuccess; }
    ~DeviceGuard() { if (prev >= 0) (void)hipSetDevice(prev); }
};

bool is_device_pointer(const void* p)
{
    hipPointerAttribute_t attr;
    if (hipPointerGetAttributes(&attr, p) != hipSuccess) { (void)hipGetLastError(); return false; }
    return attr.type == hipMemoryTypeDevice || attr.type == hipMemoryTypeManaged;
}

int grow(rt_ctx* ctx, void** buf, size_t* have, size_t need)
{
    if (*have >= need) return RT_OK;
    if (*buf) { HIP_TRY(ctx, hipFree(*buf)); *buf = nullptr; *have = 0; }
    HIP_TRY(ctx, hipMalloc(buf, need));
    *have = need;
    return RT_OK;
}

// number of refraction parents a path can stack: multiplier *= 0.9f until it is < 0.001f (renderer.h:30,118)
int refraction_chain_bound()
{
    int n = 0;
    for (float m = 1.0f; !(m < 0.001f); m = m * 0.9f) ++n;
    return n;
}

int check_params(const rt_ctx* ctx, const rt_camera35f* cam, const rt_params* p)
{
    if (!ctx || !cam || !p) return RT_ERR_INVALID_ARG;
    if (p->width <= 0 || p->height <= 0 || p->samples <= 0 || p->max_depth < 0) return RT_ERR_INVALID_ARG;
    if ((int64_t)p->width * p->height > (int64_t)1 << 30) return RT_ERR_INVALID_ARG;
    if (p->max_depth > RT_MAX_DEPTH) return RT_ERR_UNSUPPORTED;
    if (p->row0 < 0 || p->rows < 0 || p->band_rows < 0 || p->band_stride < 0) return RT_ERR_INVALID_ARG;
    if (p->rows > 0) {
        int last;
        if (p->band_rows > 0) {
            if (p->band_stride < p->band_rows) return RT_ERR_INVALID_ARG;
            const int r = p->rows - 1;
            last = p->row0 + (r / p->band_rows) * p->band_stride + (r % p->band_rows);
        } else {
            last = p->row0 + p->rows - 1;
        }
        if (last >= p->height) return RT_ERR_INVALID_ARG;
    }
    if (!ctx->has_scene) return RT_ERR_NO_SCENE;
    return RT_OK;
}

// `state`: the RT_DEV_STATE_BYTES block of the call (the context's own, or the one in the static scratch array)
void fill_args(const rt_ctx* ctx, const rt_camera35f* cam, const rt_params* p, KernelArgs& a, void* state = nullptr)
{
    unsigned long long* words = state ? (unsigned long long*)state : ctx->d_counters;
    int* fault = (int*)(words + 5);
    std::memcpy(a.cam, cam, sizeof(float) * 35);
    a.width = p->width; a.height = p->height; a.samples = p->samples; a.max_depth = p->max_depth;
    a.seed_base = (unsigned int)(uint64_t)p->seed_base;
    a.row0 = p->row0; a.rows = p->rows; a.band_rows = p->band_rows; a.band_stride = p->band_stride;
    a.tiles_x = (p->width + 15) / 16; a.tiles_y = (p->rows + 15) / 16;
    a.n_hit = ctx->n_hit; a.n_lights = ctx->n_lights; a.n_mat = ctx->n_mat;
    a.hrec = ctx->d_hrec; a.lights = ctx->d_lights; a.mats = ctx->d_mats;
    a.rgb = nullptr; a.emission = nullptr; a.counters = words; a.fault = fault;
    a.queue = (unsigned int*)(fault + 1);                // the word after it: reset with the counters
    a.keys = nullptr; a.hist = nullptr; a.list = nullptr; a.list_count = 0;
    a.fallback_count = nullptr; a.fallback_list = nullptr; a.listed_mode = 0;
    a.heavy_count = nullptr; a.heavy_rec = nullptr; a.heavy_pool = nullptr; a.heavy_pool_used = nullptr;
    a.coop_lds_stack = 0;
    a.heavy_pool_words = 0; a.heavy_cap = 0; a.heavy_min_work = 0;
    a.prim = ctx->d_prim;
    a.order = ctx->d_order;
    a.slot_of = ctx->d_slot_of;
    a.clusters = ctx->d_clusters; a.n_clusters = ctx->n_clusters;
}

// Primary rays all start at the camera: origin - centre and dot(origin, origin) - r*r (sphere.h:15,19) are per-sphere
// constants of a frame.  Same float operations, same order, done once here instead of once per ray.
int refresh_primary_terms(rt_ctx* ctx, const rt_camera35f* cam)
{
    if (ctx->nonsphere || ctx->n_hit == 0) return RT_OK;
    if (ctx->prim_valid && std::memcmp(ctx->prim_cam, cam->position, 12) == 0) return RT_OK;
    std::vector<float>& prim = ctx->h_prim;              // sized at upload
    const rtm::vec3 eye(cam->position[0], cam->position[1], cam->position[2]);
    for (int i = 0; i < ctx->n_hit; ++i) {
        const float* r = &ctx->h_centres_r2[4 * (size_t)i];
        const rtm::vec3 oc = eye - rtm::vec3(r[0], r[1], r[2]);
        prim[4 * (size_t)i + 0] = oc.x; prim[4 * (size_t)i + 1] = oc.y; prim[4 * (size_t)i + 2] = oc.z;
        prim[4 * (size_t)i + 3] = rtm::dot(oc, oc) - r[3];
    }
    HIP_TRY(ctx, hipMemcpyAsync(ctx->d_prim, prim.data(), prim.size() * sizeof(float), hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));   // `prim` is pageable memory
    std::memcpy(ctx->prim_cam, cam->position, 12);
    ctx->prim_valid = true;
    return RT_OK;
}

}  // namespace

extern "C" {

const char* rt_strerror(int status)
{
    switch (status) {
    case RT_OK:               return "ok";
    case RT_ERR_INVALID_ARG:  return "invalid argument";
    case RT_ERR_NO_DEVICE:    return "no usable gfx950 device";
    case RT_ERR_HIP:          return "HIP runtime error (see rt_last_hip_error)";
    case RT_ERR_SCENE:        return "malformed scene (material index / hittable type / light type)";
    case RT_ERR_UNSUPPORTED:  return "unsupported (reflection > 1, negative reflection / refraction, max_depth > RT_MAX_DEPTH or scene too large)";
    case RT_ERR_NO_SCENE:     return "no scene uploaded";
    case RT_ERR_DEVICE_FAULT: return "device reported a bounce-stack overflow";
    case RT_ERR_RCCL:         return "RCCL is not loadable or one of its calls failed (see rt_last_rccl_error)";
    case RT_ERR_TOO_LARGE:    return "rt_render_rows_static: rows * width exceeds RT_STATIC_MAX_PIXELS (the static scratch array holds no more)";
    default:                  return "unknown status";
    }
}

int rt_create(rt_ctx** out, int device_ordinal)
{
    if (!out) return RT_ERR_INVALID_ARG;
    *out = nullptr;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) { (void)hipGetLastError(); return RT_ERR_NO_DEVICE; }
    if (device_ordinal < 0 || device_ordinal >= ndev) return RT_ERR_NO_DEVICE;
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device_ordinal) != hipSuccess) return RT_ERR_NO_DEVICE;
    if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0) return RT_ERR_NO_DEVICE;   // the code object is gfx950 only

    rt_ctx* ctx = new (std::nothrow) rt_ctx();
    if (!ctx) return RT_ERR_INVALID_ARG;
    ctx->device = device_ordinal;
    DeviceGuard g(device_ordinal);
    if (!g.ok) { delete ctx; return RT_ERR_NO_DEVICE; }
    if (device_ordinal < RT_MAX_DEVICES) {
        std::lock_guard<std::mutex> lock(g_direct_mutex);
        ++g_direct[device_ordinal].users;
        ctx->counted = true;
    }
    hipError_t e = hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking);
    if (e == hipSuccess) e = hipEventCreate(&ctx->ev0);
    if (e == hipSuccess) e = hipEventCreate(&ctx->ev1);
    if (e == hipSuccess) e = hipEventCreate(&ctx->evm);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&ctx->ev_list, hipEventDisableTiming);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&ctx->ev_gen, hipEventDisableTiming);
    if (e == hipSuccess) e = hipEventCreate(&ctx->ev_l0);
    if (e == hipSuccess) e = hipEventCreate(&ctx->ev_l1);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&ctx->ev_wait, hipEventDisableTiming);
    if (e == hipSuccess) e = hipEventCreate(&ctx->ev_g0);
    if (e == hipSuccess) e = hipEventCreate(&ctx->ev_g1);
    if (e == hipSuccess) e = hipHostMalloc((void**)&ctx->h_back, 8 * sizeof(unsigned long long), hipHostMallocDefault);
    // (stream2 / stream3 are created when a frame first needs them: HIP deals streams to its hardware queues as they are
    // created, and callers with many contexts in flight -- RT_FLAG_SEQUENCE -- should not have idle streams pushing
    // busy ones onto shared queues)
    if (e == hipSuccess) e = hipEventCreateWithFlags(&ctx->ev_rest, hipEventDisableTiming);
    if (e == hipSuccess) e = hipMalloc((void**)&ctx->d_counters, RT_DEV_STATE_BYTES);
    if (e == hipSuccess && device_ordinal < RT_MAX_DEVICES) {
        // once per device, with that device current (DeviceGuard above): this GPU's address of the scratch array
        std::lock_guard<std::mutex> lock(g_scratch_mutex);
        if (!g_scratch_ptr[device_ordinal]) {
            unsigned char* ptr = nullptr;
            e = hipGetSymbolAddress((void**)&ptr, HIP_SYMBOL(scratch));
            if (e == hipSuccess) g_scratch_ptr[device_ordinal] = ptr;
        }
        ctx->scratch = g_scratch_ptr[device_ordinal];
    }
    if (e != hipSuccess) { ctx->last_hip = (int)e; rt_destroy(ctx); return RT_ERR_HIP; }
    ctx->d_fault = (int*)(ctx->d_counters + 5);
    *out = ctx;
    return RT_OK;
}

void rt_destroy(rt_ctx* ctx)
{
    if (!ctx) return;
    DeviceGuard g(ctx->device);
    if (ctx->stream) (void)hipStreamSynchronize(ctx->stream);
    if (ctx->d_scene) (void)hipFree(ctx->d_scene);
    if (ctx->d_counters) (void)hipFree(ctx->d_counters);
    if (ctx->d_stage_rgb) (void)hipFree(ctx->d_stage_rgb);
    if (ctx->d_stage_em) (void)hipFree(ctx->d_stage_em);
    if (ctx->ev0) (void)hipEventDestroy(ctx->ev0);
    if (ctx->ev1) (void)hipEventDestroy(ctx->ev1);
    if (ctx->evm) (void)hipEventDestroy(ctx->evm);
    if (ctx->ev_list) (void)hipEventDestroy(ctx->ev_list);
    if (ctx->ev_gen) (void)hipEventDestroy(ctx->ev_gen);
    if (ctx->ev_l0) (void)hipEventDestroy(ctx->ev_l0);
    if (ctx->ev_l1) (void)hipEventDestroy(ctx->ev_l1);
    if (ctx->ev_wait) (void)hipEventDestroy(ctx->ev_wait);
    if (ctx->ev_g0) (void)hipEventDestroy(ctx->ev_g0);
    if (ctx->ev_g1) (void)hipEventDestroy(ctx->ev_g1);
    if (ctx->h_back) (void)hipHostFree(ctx->h_back);
    if (ctx->stream2) { (void)hipStreamSynchronize(ctx->stream2); (void)hipStreamDestroy(ctx->stream2); }
    if (ctx->ev_rest) (void)hipEventDestroy(ctx->ev_rest);
    if (ctx->stream3) { (void)hipStreamSynchronize(ctx->stream3); (void)hipStreamDestroy(ctx->stream3); }
    if (ctx->d_classes) (void)hipFree(ctx->d_classes);
    if (ctx->d_heavy) (void)hipFree(ctx->d_heavy);
    if (ctx->d_tiles) (void)hipFree(ctx->d_tiles);
    if (ctx->d_gather) (void)hipFree(ctx->d_gather);
    if (ctx->d_glow) (void)hipFree(ctx->d_glow);
    if (ctx->stream) (void)hipStreamDestroy(ctx->stream);
    if (ctx->counted) {
        // the last context of a device takes the shared streams with it (idle: every context waited for its own kernels)
        std::lock_guard<std::mutex> lock(g_direct_mutex);
        SharedDirect& sd = g_direct[ctx->device];
        if (--sd.users == 0)
            for (hipStream_t& st : sd.s)
                if (st) { (void)hipStreamSynchronize(st); (void)hipStreamDestroy(st); st = nullptr; }
    }
    delete ctx;
}

int rt_last_hip_error(const rt_ctx* ctx) { return ctx ? ctx->last_hip : 0; }
float rt_last_kernel_ms(const rt_ctx* ctx) { return ctx ? ctx->last_ms : 0.0f; }
float rt_last_device_ms(const rt_ctx* ctx) { return ctx ? ctx->last_total_ms : 0.0f; }
float rt_last_gather_ms(const rt_ctx* ctx) { return ctx ? ctx->last_gather_ms : 0.0f; }

int rt_camera_init(rt_camera35f* cam, float fov_deg, float width, float height, float z_near, float z_far)
{
    if (!cam || !(width > 0.0f) || !(height > 0.0f) || !(fov_deg > 0.0f)) return RT_ERR_INVALID_ARG;
    Camera c(fov_deg, width, height, z_near, z_far);
    cam->position[0] = c.GetPosition().x; cam->position[1] = c.GetPosition().y; cam->position[2] = c.GetPosition().z;
    std::memcpy(cam->inv_proj, c.GetInverseProjectionMatrix().m, sizeof(float) * 16);
    std::memcpy(cam->inv_view, c.GetInverseViewMatrix().m, sizeof(float) * 16);
    return RT_OK;
}

int rt_upload_scene(rt_ctx* ctx, const rt_scene_soa* s)
{
    if (!ctx || !s) return RT_ERR_INVALID_ARG;
    if (s->n_materials < 0 || s->n_hittables < 0 || s->n_lights < 0) return RT_ERR_INVALID_ARG;
    if ((s->n_materials > 0 && !s->materials) || (s->n_lights > 0 && (!s->l_type || !s->l_v)) ||
        (s->n_hittables > 0 && (!s->h_type || !s->h_a || !s->h_b || !s->h_radius || !s->h_material)))
        return RT_ERR_INVALID_ARG;
    if (s->n_hittables > RT_MAX_HITTABLES || s->n_lights > RT_MAX_LIGHTS || s->n_materials > RT_MAX_MATERIALS)
        return RT_ERR_UNSUPPORTED;

    bool mixed = false, any_bounce = false;
    for (int i = 0; i < s->n_materials; ++i) {
        const float reflection = s->materials[10 * i + 4], refraction = s->materials[10 * i + 5];
        // reflection > 1 lets the multiplier grow: the reference's recursion is then bounded by nothing but depth
        // for reflections and by nothing at all through refractions (renderer.h:82,118)
        if (reflection > 1.0f || reflection != reflection || refraction != refraction) return RT_ERR_UNSUPPORTED;
        // a negative reflection or refraction takes neither branch of renderer.h:74,93 and fails the `both zero` test of
        // :128, so maxDepth stays 0 and AntiAliasing divides by it (:169): inf / NaN pixels in the reference.  The
        // kernel pair keys such a material as direct (maxDepth 1): refused rather than rendered differently.
        if (reflection < 0.0f || refraction < 0.0f) return RT_ERR_UNSUPPORTED;
        if (reflection > 0.0f && refraction > 0.0f) mixed = true;
        if (reflection > 0.0f || refraction > 0.0f) any_bounce = true;
    }

    const int nh = s->n_hittables, nl = s->n_lights, nm = s->n_materials;
    // hrec | lights | mats (padded to 4 floats) | prim (filled per camera) | sorted order of the hittables | its inverse |
    // the clusters' boxes, 8 floats each
    const size_t nclusters = ((size_t)nh + 63) / 64;
    std::vector<float> img((size_t)nh * 8 + (size_t)nl * 4 + (size_t)nm * 10 + 4 + (size_t)nh * 4 + 2 * (size_t)nh + 8 * nclusters, 0.0f);
    float* hrec = img.data();
    float* lights = hrec + (size_t)nh * 8;
    float* mats = lights + (size_t)nl * 4;
    bool nonsphere = false;
    for (int i = 0; i < nh; ++i) {
        const int type = s->h_type[i];
        int material = s->h_material[i];
        const float* a = s->h_a + 3 * i; const float* b = s->h_b + 3 * i;
        float* r = hrec + 8 * i;
        r[0] = a[0]; r[1] = a[1]; r[2] = a[2];
        if (type == RT_SPHERE) {
            material = (uint8_t)material;                                   // sphere.h:10 takes uint8_t
            r[3] = s->h_radius[i] * s->h_radius[i];                         // sphere.h:19
        } else if (type == RT_PLANE) {
            material = (uint8_t)material;                                   // plane.h:10
            const rtm::vec3 n = rtm::normalize(rtm::vec3(b[0], b[1], b[2])); // plane.h:11
            r[4] = n.x; r[5] = n.y; r[6] = n.z;
            nonsphere = true;
        } else if (type == RT_CUBE) {
            r[4] = a[0] + b[0]; r[5] = a[1] + b[1]; r[6] = a[2] + b[2];     // cube.h:17,33,37
            nonsphere = true;
        } else {
            return RT_ERR_SCENE;
        }
        if (material < 0 || material >= nm) return RT_ERR_SCENE;
        const int bits = type | (material << 2);
        std::memcpy(&r[7], &bits, 4);
    }
    for (int i = 0; i < nl; ++i) {
        const int type = s->l_type[i];
        const float* v = s->l_v + 3 * i;
        float* r = lights + 4 * i;
        if (type == RT_DIRLIGHT) {
            const rtm::vec3 d = rtm::normalize(-rtm::vec3(v[0], v[1], v[2]));  // directionalLight.h:9
            r[0] = d.x; r[1] = d.y; r[2] = d.z;
        } else if (type == RT_POINTLIGHT) {
            r[0] = v[0]; r[1] = v[1]; r[2] = v[2];
        } else {
            return RT_ERR_SCENE;
        }
        std::memcpy(&r[3], &type, 4);
    }
    if (nm > 0) std::memcpy(mats, s->materials, sizeof(float) * 10 * (size_t)nm);
    // Cooperative scans hand hittables to lanes in this order: same type side by side, so the 64 tests of a round mostly
    // take the same branch, and within a type in tiles, so the 64 hittables of a round -- a *cluster* -- sit together
    // in space (the winner is picked by index, so the order of testing does not matter).  Per cluster an axis-aligned
    // box (centre, half extents): a query whose ray certainly misses it skips the round (rt_device.hpp cluster_mask has
    // the argument).  Planes and hittables with non-finite parameters make their cluster's box infinite: always
    // visited.  Hittables much larger than the rest (a ground sphere) go to the end of their type, so that they spoil
    // one cluster's box and not the tiling.
    const size_t prim_at = (size_t)nh * 8 + (size_t)nl * 4 + (((size_t)nm * 10 + 3) & ~(size_t)3);
    const size_t clusters_at = prim_at + (size_t)nh * 6;
    {
        // every hittable's box: spheres centre -/+ |radius| (sphere.h:19 squares the radius), cubes the two corners of
        // cube.h:16-17 either way round (cube.h:19-20 sorts the slabs' ends)
        std::vector<double> blo((size_t)nh * 3), bhi((size_t)nh * 3), ext((size_t)nh);
        for (int i = 0; i < nh; ++i) {
            const float* r = hrec + 8 * (size_t)i;
            const int type = s->h_type[i];
            double* l = &blo[(size_t)i * 3]; double* h = &bhi[(size_t)i * 3];
            bool bounded = type == RT_SPHERE || type == RT_CUBE;
            for (int k = 0; k < 3 && bounded; ++k) {
                if (type == RT_SPHERE) { const double rad = std::fabs((double)s->h_radius[i]); l[k] = (double)r[k] - rad; h[k] = (double)r[k] + rad; }
                else { const double p = r[k], q = r[4 + k]; l[k] = p < q ? p : q; h[k] = p < q ? q : p; }
                bounded = std::isfinite(l[k]) && std::isfinite(h[k]);
            }
            double d2 = 0.0;
            for (int k = 0; k < 3 && bounded; ++k) d2 += (h[k] - l[k]) * (h[k] - l[k]);
            ext[(size_t)i] = bounded ? std::sqrt(d2) : INFINITY;
        }
        std::vector<double> sizes;
        for (int i = 0; i < nh; ++i) if (std::isfinite(ext[(size_t)i])) sizes.push_back(ext[(size_t)i]);
        double big = INFINITY;
        if (!sizes.empty()) { std::nth_element(sizes.begin(), sizes.begin() + (long)(sizes.size() / 2), sizes.end()); big = 8.0 * sizes[sizes.size() / 2]; }
        auto centre = [&](uint32_t i, int k) { return 0.5 * (blo[(size_t)i * 3 + k] + bhi[(size_t)i * 3 + k]); };
        // tiles: order[from, to) is cut at the round boundary (a multiple of 64) nearest its middle, across the axis
        // along which its centres spread most, until every piece lies within one round
        std::vector<uint32_t> order;
        order.reserve((size_t)nh);
        struct Range { size_t from, to; };
        std::vector<Range> todo;
        auto tile = [&](size_t from0, size_t to0) {
            todo.push_back({ from0, to0 });
            while (!todo.empty()) {
                const Range g = todo.back(); todo.pop_back();
                if (g.to <= g.from || g.from / 64 == (g.to - 1) / 64) continue;
                double lo[3] = { 1e300, 1e300, 1e300 }, hi[3] = { -1e300, -1e300, -1e300 };
                for (size_t j = g.from; j < g.to; ++j)
                    for (int k = 0; k < 3; ++k) { const double c = centre(order[j], k); lo[k] = c < lo[k] ? c : lo[k]; hi[k] = c > hi[k] ? c : hi[k]; }
                int axis = 0;
                for (int k = 1; k < 3; ++k) if (hi[k] - lo[k] > hi[axis] - lo[axis]) axis = k;
                std::stable_sort(order.begin() + (long)g.from, order.begin() + (long)g.to,
                                 [&](uint32_t x, uint32_t y) { return centre(x, axis) < centre(y, axis); });
                const size_t first = g.from / 64 + 1, last = (g.to - 1) / 64;            // boundaries first*64 .. last*64 lie inside
                const size_t mid = 64 * (first + (last - first) / 2);
                todo.push_back({ g.from, mid });
                todo.push_back({ mid, g.to });
            }
        };
        for (int type = RT_SPHERE; type <= RT_CUBE; ++type) {
            const size_t from = order.size();
            for (int i = 0; i < nh; ++i) if (s->h_type[i] == type && ext[(size_t)i] <= big) order.push_back((uint32_t)i);
            tile(from, order.size());
            for (int i = 0; i < nh; ++i) if (s->h_type[i] == type && std::isfinite(ext[(size_t)i]) && !(ext[(size_t)i] <= big)) order.push_back((uint32_t)i);
            for (int i = 0; i < nh; ++i) if (s->h_type[i] == type && !std::isfinite(ext[(size_t)i])) order.push_back((uint32_t)i);
        }
        if (nh > 0) std::memcpy(&img[prim_at + (size_t)nh * 4], order.data(), sizeof(uint32_t) * (size_t)nh);
        // ... and slot_of[hittable] = its place in that order (the sequential fallback walks the scene in scene order)
        std::vector<uint32_t> slot_of((size_t)nh);
        for (int j = 0; j < nh; ++j) slot_of[order[(size_t)j]] = (uint32_t)j;
        if (nh > 0) std::memcpy(&img[prim_at + (size_t)nh * 5], slot_of.data(), sizeof(uint32_t) * (size_t)nh);
        // the clusters' boxes in float, rounded outwards: (centre.xyz, 1 if a cube is among the members: cluster_mask
        // is more careful with those), (half.xyz, 0)
        float* cl = nclusters > 0 ? &img[clusters_at] : nullptr;
        for (size_t k = 0; k < nclusters; ++k) {
            const size_t from = 64 * k, to = from + 64 < (size_t)nh ? from + 64 : (size_t)nh;
            double l[3] = { 1e300, 1e300, 1e300 }, h[3] = { -1e300, -1e300, -1e300 };
            bool open = false, cubes = false;
            for (size_t j = from; j < to; ++j) {
                const uint32_t i = order[j];
                if (!std::isfinite(ext[i])) { open = true; continue; }
                if (s->h_type[i] != RT_SPHERE) cubes = true;
                for (int d = 0; d < 3; ++d) { l[d] = blo[(size_t)i * 3 + d] < l[d] ? blo[(size_t)i * 3 + d] : l[d]; h[d] = bhi[(size_t)i * 3 + d] > h[d] ? bhi[(size_t)i * 3 + d] : h[d]; }
            }
            float* c = cl + 8 * k;
            for (int d = 0; d < 3; ++d) {
                const float cf = open ? 0.0f : (float)(0.5 * (l[d] + h[d]));
                const double need = open ? (double)INFINITY : std::fmax(h[d] - (double)cf, (double)cf - l[d]);   // covers [l, h] around cf
                float hf = (float)need;
                if ((double)hf < need) hf = std::nextafterf(hf, INFINITY);
                c[d] = cf; c[4 + d] = hf;
                if (!std::isfinite(cf) || !std::isfinite(hf)) { c[d] = 0.0f; c[4 + d] = INFINITY; }
            }
            c[3] = cubes ? 1.0f : 0.0f;
        }
    }

    if (rt_launch_lds_bytes(nh, nl, nm) > 160 * 1024) return RT_ERR_UNSUPPORTED;

    DeviceGuard g(ctx->device);
    if (!g.ok) return RT_ERR_NO_DEVICE;
    const size_t bytes = img.size() * sizeof(float);
    if (bytes > ctx->scene_bytes) {
        if (ctx->d_scene) { HIP_TRY(ctx, hipFree(ctx->d_scene)); ctx->d_scene = nullptr; ctx->scene_bytes = 0; }
        HIP_TRY(ctx, hipMalloc(&ctx->d_scene, bytes));
        ctx->scene_bytes = bytes;
    }
    ctx->has_scene = false;
    ctx->listed_timed = false; ctx->listed_ms = 0.0f;
    HIP_TRY(ctx, hipMemcpyAsync(ctx->d_scene, img.data(), bytes, hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    ctx->d_hrec = (const float4*)ctx->d_scene;
    ctx->d_lights = (const float4*)((const float*)ctx->d_scene + (size_t)nh * 8);
    ctx->d_mats = (const float*)ctx->d_scene + (size_t)nh * 8 + (size_t)nl * 4;
    ctx->d_prim = (float4*)((float*)ctx->d_scene + prim_at);
    ctx->d_order = (const unsigned int*)((const float*)ctx->d_scene + prim_at + (size_t)nh * 4);
    ctx->d_slot_of = ctx->d_order + nh;
    ctx->d_clusters = (const float*)ctx->d_scene + clusters_at;
    ctx->n_clusters = (int)nclusters;
    ctx->h_centres_r2.resize((size_t)nh * 4);
    ctx->h_prim.assign((size_t)nh * 4, 0.0f);
    for (int i = 0; i < nh; ++i) std::memcpy(&ctx->h_centres_r2[4 * (size_t)i], hrec + 8 * (size_t)i, 16);
    ctx->prim_valid = false;
    ctx->n_hit = nh; ctx->n_lights = nl; ctx->n_mat = nm;
    ctx->mixed = mixed; ctx->nonsphere = nonsphere; ctx->any_bounce = any_bounce;
    ctx->has_scene = true;
    return RT_OK;
}

int rt_debug_cull(rt_ctx* ctx, int32_t index, const float* rays, int32_t n_rays, uint8_t* out)
{
    if (!ctx || !rays || !out || n_rays <= 0) return RT_ERR_INVALID_ARG;
    if (!ctx->has_scene) return RT_ERR_NO_SCENE;
    if (index < 0 || index >= ctx->n_hit) return RT_ERR_INVALID_ARG;
    DeviceGuard g(ctx->device);
    if (!g.ok) return RT_ERR_NO_DEVICE;
    float* d_rays = nullptr; unsigned char* d_out = nullptr;
    HIP_TRY(ctx, hipMalloc((void**)&d_rays, sizeof(float) * 6 * (size_t)n_rays));
    hipError_t e = hipMalloc((void**)&d_out, (size_t)n_rays);
    if (e == hipSuccess) e = hipMemcpyAsync(d_rays, rays, sizeof(float) * 6 * (size_t)n_rays, hipMemcpyHostToDevice, ctx->stream);
    if (e == hipSuccess) e = rt_launch_debug_cull(ctx->d_clusters, ctx->d_slot_of, index, d_rays, n_rays, d_out, ctx->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(out, d_out, (size_t)n_rays, hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    (void)hipFree(d_rays); if (d_out) (void)hipFree(d_out);
    HIP_TRY(ctx, e);
    return RT_OK;
}

int rt_debug_cull_enter(rt_ctx* ctx, int32_t index, const float* rays, int32_t n_rays, float* enter)
{
    if (!ctx || !rays || !enter || n_rays <= 0) return RT_ERR_INVALID_ARG;
    if (!ctx->has_scene) return RT_ERR_NO_SCENE;
    if (index < 0 || index >= ctx->n_hit) return RT_ERR_INVALID_ARG;
    DeviceGuard g(ctx->device);
    if (!g.ok) return RT_ERR_NO_DEVICE;
    float* d_rays = nullptr; float* d_out = nullptr;
    HIP_TRY(ctx, hipMalloc((void**)&d_rays, sizeof(float) * 6 * (size_t)n_rays));
    hipError_t e = hipMalloc((void**)&d_out, sizeof(float) * (size_t)n_rays);
    if (e == hipSuccess) e = hipMemcpyAsync(d_rays, rays, sizeof(float) * 6 * (size_t)n_rays, hipMemcpyHostToDevice, ctx->stream);
    if (e == hipSuccess) e = rt_launch_debug_cull_enter(ctx->d_clusters, ctx->d_slot_of, index, d_rays, n_rays, d_out, ctx->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(enter, d_out, sizeof(float) * (size_t)n_rays, hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    (void)hipFree(d_rays); if (d_out) (void)hipFree(d_out);
    HIP_TRY(ctx, e);
    return RT_OK;
}

}  // extern "C"

// rt_render_rows and rt_render_rows_static.  `no_alloc`: the call may not allocate -- its per-frame state and pixel
// classes live in this GPU's copy of the static scratch array (ctx->scratch), outputs must be device pointers, no
// tallies, and the sample-parallel route (which needs a stream pool) is left out: long pixels stay with the listed
// kernel, same bytes.
// The host waits for the context's stream.  hipStreamSynchronize polls without pause (HIP's default); a caller with frames
// in flight (RT_FLAG_SEQUENCE) has a host thread per frame, and 16 of them polling kept 12 CPUs busy.  Those callers
// poll an event with short sleeps in between instead: the wake-up is late by up to the sleep, which the other frames in
// flight hide (DESIGN.md section 6 has the measurement; HIP's own sleeping waits are a process-wide device flag and
// not the library's to set).
#ifndef RT_WAIT_SLEEP_US
#define RT_WAIT_SLEEP_US 40
#endif
static hipError_t wait_stream(rt_ctx* ctx, bool sleep)
{
    if (!sleep) return hipStreamSynchronize(ctx->stream);
    static const long sleep_us = getenv("RT_WAIT_SLEEP_US") ? std::atol(getenv("RT_WAIT_SLEEP_US")) : RT_WAIT_SLEEP_US;
    if (sleep_us <= 0) return hipStreamSynchronize(ctx->stream);
    hipError_t e = hipEventRecord(ctx->ev_wait, ctx->stream);
    if (e != hipSuccess) return e;
    const timespec nap = { 0, sleep_us * 1000 };
    bool polled = false;
    while ((e = hipEventQuery(ctx->ev_wait)) == hipErrorNotReady) { polled = true; nanosleep(&nap, nullptr); }
    // hipErrorNotReady is an answer, not a failure, but HIP keeps it as the thread's last error: take it back, or the
    // caller's next hipGetLastError() (torch checks it after every kernel launch) reports "device not ready"
    if (polled && e == hipSuccess) (void)hipGetLastError();
    return e;
}

static int render_rows_impl(rt_ctx* ctx, const rt_camera35f* cam, const rt_params* p,
                            uint8_t* rgb_out, float* emission_out, rt_counters* counters, bool no_alloc)
{
    int rc = check_params(ctx, cam, p);
    if (rc != RT_OK) return rc;
    const bool count = (p->flags & (RT_FLAG_COUNT_LIKE_REFERENCE | RT_FLAG_COUNT_EXECUTED)) != 0;
    if (count && !counters) return RT_ERR_INVALID_ARG;
    if (!rgb_out && !emission_out && !count) return RT_ERR_INVALID_ARG;
    if (counters) std::memset(counters, 0, sizeof(*counters));
    if (p->rows == 0) { ctx->last_ms = 0.0f; ctx->last_total_ms = 0.0f; return RT_OK; }
    const size_t npix = (size_t)p->width * p->rows;
    if (no_alloc) {
        if (count || counters) return RT_ERR_INVALID_ARG;
        if (!ctx->scratch) return RT_ERR_NO_DEVICE;
        if (npix > (size_t)RT_STATIC_MAX_PIXELS) return RT_ERR_TOO_LARGE;
    }

    DeviceGuard g(ctx->device);
    if (!g.ok) return RT_ERR_NO_DEVICE;

    const bool rgb_dev = rgb_out && is_device_pointer(rgb_out);
    const bool em_dev = emission_out && is_device_pointer(emission_out);
    if (no_alloc && ((rgb_out && !rgb_dev) || (emission_out && !em_dev))) return RT_ERR_INVALID_ARG;   // host outputs need staging
    rc = refresh_primary_terms(ctx, cam);
    if (rc != RT_OK) return rc;
    // the call's RT_DEV_STATE_BYTES block: tallies, fault flag, queue word, hand-over counters, histogram + cursor
    unsigned char* const state = no_alloc ? ctx->scratch : (unsigned char*)ctx->d_counters;
    KernelArgs a;
    fill_args(ctx, cam, p, a, state);
    if (rgb_out) {
        if (rgb_dev) a.rgb = rgb_out;
        else { rc = grow(ctx, &ctx->d_stage_rgb, &ctx->stage_rgb_bytes, npix * 3); if (rc) return rc; a.rgb = (unsigned char*)ctx->d_stage_rgb; }
    }
    if (emission_out) {
        if (em_dev) a.emission = (float4*)emission_out;
        else { rc = grow(ctx, &ctx->d_stage_em, &ctx->stage_em_bytes, npix * 16); if (rc) return rc; a.emission = (float4*)ctx->d_stage_em; }
    }

    LaunchTraits t;
    t.emission = emission_out != nullptr;
    t.mixed = ctx->mixed;
    t.nonsphere = ctx->nonsphere;
    t.count = count;
    t.listed = false;
    t.rng_scratch = false;
    t.wave_groups = false;
    t.prune = (p->flags & RT_FLAG_COUNT_LIKE_REFERENCE) == 0;
    // frames on the bounce stack <= reflection parents (max_depth) + refraction parents (66)
    t.deep = refraction_chain_bound() + p->max_depth > RT_STACK_SHALLOW || p->max_depth > RT_PEND_SHALLOW - 1;

    // (the scratch array keeps the last call's contents: everything that must start at zero is in this block)
    HIP_TRY(ctx, hipMemsetAsync(state, 0, RT_DEV_STATE_BYTES, ctx->stream));
    HIP_TRY(ctx, hipEventRecord(ctx->ev0, ctx->stream));
    if (count || (p->flags & RT_FLAG_SINGLE_KERNEL)) {
        // tallying builds and the A/B flag: the general kernel over every pixel
        HIP_TRY(ctx, hipEventRecord(ctx->evm, ctx->stream));
        HIP_TRY(ctx, rt_launch_render(a, t, ctx->stream));
    } else {
        // Two kernels side by side (DESIGN.md "execution model").  Pixels whose primary rays can bounce -- found by
        // one closest-hit query per primary direction -- go, grouped by material, to the general kernel on a second
        // stream; all other pixels need no RNG, stack or scratch and go to the direct kernel.
        const bool bounces = ctx->any_bounce && p->max_depth >= 1;       // renderer.h:68: depth 1 <= maxTotalDepth
        // RT_FLAG_RELAXED: the direct kernel of rt_kernels_relaxed.hip (fused multiply-adds, 1-ulp sqrt / reciprocal)
        const bool relaxed = (p->flags & RT_FLAG_RELAXED) != 0;
        auto launch_direct = [relaxed](const KernelArgs& ka, const LaunchTraits& lt, hipStream_t st) {
            return relaxed ? rt_launch_direct_relaxed(ka, lt, st) : rt_launch_direct(ka, lt, st);
        };
        unsigned int n_listed = 0;
        bool rest = false, second = false, timed = false, direct_done = false;
        if (bounces) {
            unsigned int* list;
            if (no_alloc) {
                a.keys = (unsigned short*)(ctx->scratch + RT_STATIC_STATE_BYTES);
                list = (unsigned int*)(ctx->scratch + RT_STATIC_STATE_BYTES + RT_STATIC_KEYS_BYTES);
            } else {
                const size_t keys_bytes = (npix * 2 + 15) & ~(size_t)15;
                rc = grow(ctx, &ctx->d_classes, &ctx->classes_bytes, keys_bytes + npix * 4);
                if (rc) return rc;
                a.keys = (unsigned short*)ctx->d_classes;
                list = (unsigned int*)((char*)ctx->d_classes + keys_bytes);
            }
            a.hist = (unsigned int*)(state + RT_DEV_HIST_AT);                         // zeroed with the counters
            unsigned int* cursor = a.hist + RT_MAX_KEYS;
            HIP_TRY(ctx, rt_launch_classify(a, t, cursor, list, ctx->stream));
            HIP_TRY(ctx, hipEventRecord(ctx->ev_list, ctx->stream));
            // the list length sizes the general kernel's grid: scan_kernel leaves it in cursor[0] ... which scatter does
            // not touch (keys are >= 1), so it can be read as soon as the list is complete
            HIP_TRY(ctx, hipMemcpyAsync(&ctx->h_back[6], cursor, sizeof(n_listed), hipMemcpyDeviceToHost, ctx->stream));
            HIP_TRY(ctx, wait_stream(ctx, (p->flags & RT_FLAG_SEQUENCE) != 0));
            std::memcpy(&n_listed, &ctx->h_back[6], sizeof(n_listed));
            if (n_listed > 0) {
                KernelArgs g2 = a;
                g2.list = list; g2.list_count = (int)n_listed;
                LaunchTraits tg = t;
                tg.listed = true;
                // Where the listed kernel keeps its generators (31 words per pixel).  In LDS a lone wave is ~5 % faster,
                // which is what a frame bound by its longest pixels wants; in scratch a listed workgroup holds 12 KB of
                // LDS instead of 44 KB, which is what the direct kernel's workgroups want when the frame is bound by
                // them (4K / 64 spp: 5.83 -> 5.73 ms) and what a sequence of frames wants (6 in flight: 4K frame 5.59 ->
                // 5.39 ms, glass 1080p 65 -> 51 ms per frame; measured A/B on one box).
                const double direct_work = (double)(npix - n_listed) * (double)p->samples;
                bool no_handover = false;                  // the listed kernel keeps every pixel: no heavy / resume launch
                tg.rng_scratch = (p->flags & RT_FLAG_SEQUENCE) != 0 || direct_work >= RT_DIRECT_BOUND_WORK;
                // Long pixels leave the listed kernel after their first sample and are finished one wave per pixel,
                // one lane per sample, by the heavy kernel behind it (DESIGN.md "sample-parallel pixels").
                const bool force = (p->flags & RT_FLAG_FORCE_SAMPLE_PARALLEL) != 0;
                // Large scenes: one wave per listed pixel, the lanes share every scan of the scene (coop_kernel)
                const bool coop = ctx->n_hit >= RT_COOP_MIN_HIT && rt_launch_coop_fits(ctx->n_hit, ctx->n_lights, ctx->n_mat)
                                  && !force;
                if (!coop && !no_alloc && p->samples >= 8 && p->samples <= 64 && rt_launch_heavy_fits(ctx->n_hit, ctx->n_lights, ctx->n_mat)) {
                    // records + a pool for the pixels' written-out rand() sequences
                    size_t cap = n_listed < RT_HEAVY_CAP ? n_listed : RT_HEAVY_CAP;
                    // the pool follows the list: 64 bytes per listed pixel and sample (a pixel that draws 8 numbers per
                    // sample fills a quarter of that), between 16 MiB and RT_HEAVY_TABLE_BYTES; a full pool only means
                    // that further long pixels stay serial.  Grown, never shrunk: a sequence of frames allocates once.
                    size_t pool_bytes = (size_t)n_listed * (size_t)p->samples * 64;
                    if (pool_bytes < ((size_t)16 << 20)) pool_bytes = (size_t)16 << 20;
                    if (pool_bytes > RT_HEAVY_TABLE_BYTES) pool_bytes = RT_HEAVY_TABLE_BYTES;
                    if (ctx->heavy_pool_bytes > pool_bytes) pool_bytes = ctx->heavy_pool_bytes;
                    ctx->heavy_pool_bytes = pool_bytes;
                    const size_t pool_words = pool_bytes / 4;
                    rc = grow(ctx, &ctx->d_heavy, &ctx->heavy_bytes, 16 + 4 * (pool_words + cap * (RT_HEAVY_REC_WORDS + 2)));
                    if (rc) return rc;
                    g2.heavy_count = (unsigned int*)(state + RT_DEV_HEAVY_AT);   // zeroed with the counters
                    g2.fallback_count = g2.heavy_count + 1;
                    g2.heavy_pool_used = (unsigned long long*)(g2.heavy_count + 2);
                    g2.heavy_pool = (unsigned int*)ctx->d_heavy;
                    g2.heavy_pool_words = (unsigned int)pool_words;
                    g2.heavy_rec = g2.heavy_pool + pool_words;
                    g2.fallback_list = g2.heavy_rec + cap * RT_HEAVY_REC_WORDS;
                    g2.heavy_cap = (int)cap;
                    // the threshold is tuned on DEFAULT-4 (4 hittables, 1 light: 8 primitive tests per query pair);
                    // a bounce costs in proportion to hittables x (1 + lights)
                    const double per_bounce = (double)ctx->n_hit * (1.0 + ctx->n_lights) / 8.0;
                    static const double seq_min_work =
                        getenv("RT_HEAVY_MIN_WORK_SEQ") ? atof(getenv("RT_HEAVY_MIN_WORK_SEQ")) : (double)RT_HEAVY_MIN_WORK_SEQ;
                    double min_work = ((p->flags & RT_FLAG_SEQUENCE) && seq_min_work >= 1.0 ? seq_min_work : (double)RT_HEAVY_MIN_WORK)
                                      / (per_bounce > 1.0 ? per_bounce : 1.0);
                    g2.heavy_min_work = force ? 0u : (unsigned int)(min_work < 1.0 ? 1.0 : min_work > 4.0e9 ? 4.0e9 : min_work);
                    // With frames in flight the hand-over pays where the frame's direct kernel keeps the GPU busy long
                    // enough: its rounds trace some samples twice, which is free beside a long direct kernel and shortens
                    // the frame's tail, and costs throughput where other frames' kernels would have used the slots
                    // (measured by share of the 4K frame and on the glass frame).  Below the bar that
                    // rng_scratch uses for "bound by its direct kernel" nothing is handed over (and nothing launched for it).
                    if ((p->flags & RT_FLAG_SEQUENCE) && direct_work < RT_DIRECT_BOUND_WORK && !force && !getenv("RT_HEAVY_MIN_WORK_SEQ")) {
                        g2.heavy_min_work = 0xFFFFFFFFu;
                        no_handover = true;
                    }
                }
                // Frames in flight (RT_FLAG_SEQUENCE): the listed kernels go onto the context's one stream (the direct
                // kernel onto one of the device's shared streams, below; behind the listed kernels on this stream when
                // those are off) -- the kernels of the other frames in flight run beside them.  HIP deals streams to
                // hardware queues as they are created; a context that never needs a second stream never creates one, so
                // N contexts in flight are N busy queues plus the shared ones.
                static const int seq_streams = getenv("RT_SEQ_STREAMS") ? atoi(getenv("RT_SEQ_STREAMS")) : RT_SEQ_STREAMS_DEFAULT;
                const bool sequence = g2.heavy_cap > 0 && !coop && (p->flags & RT_FLAG_SEQUENCE) != 0;
                hipStream_t s2 = ctx->stream;
                // (rt_render_rows_static creates nothing, streams included: its listed kernel runs in front of the direct one)
                // Large scenes: the direct kernel (tens of ms) runs in front of coop_kernel on the same stream.  Side by
                // side they take the same time when coop_kernel's workgroups are placed first -- and half as long again
                // when the direct kernel's are (measured: 1.18 s instead of 0.81 s): a coop workgroup's 12 waves then land
                // on whichever SIMDs have registers left, and stay there for the whole frame.
                if (!(sequence && seq_streams == 1) && !no_alloc && !coop) {
                    if (!ctx->stream2) HIP_TRY(ctx, hipStreamCreateWithFlags(&ctx->stream2, hipStreamNonBlocking));
                    s2 = ctx->stream2;
                    HIP_TRY(ctx, hipStreamWaitEvent(s2, ctx->ev_list, 0));
                }
                if (s2 == ctx->stream) { HIP_TRY(ctx, hipEventRecord(ctx->evm, ctx->stream)); timed = true; }   // the kernel span starts here
                if (coop) {
                    if (n_listed < npix) { HIP_TRY(ctx, launch_direct(a, t, ctx->stream)); direct_done = true; }
                    HIP_TRY(ctx, rt_launch_coop(g2, tg, s2));
                } else if (sequence) {
                    // a long listed kernel (this context's last frame of the scene says): one wave per workgroup
                    // (the last frame of this context has been waited for: its events are complete)
                    static const double wave_groups_ms = getenv("RT_WAVE_GROUPS_MS") ? atof(getenv("RT_WAVE_GROUPS_MS")) : RT_WAVE_GROUPS_MS;
                    if (ctx->listed_timed) {
                        float ms = 0.0f;
                        if (hipEventElapsedTime(&ms, ctx->ev_l0, ctx->ev_l1) == hipSuccess) ctx->listed_ms = ms;
                        else (void)hipGetLastError();
                        ctx->listed_timed = false;
                    }
                    tg.wave_groups = (p->flags & RT_FLAG_FORCE_WAVE_GROUPS) != 0 || wave_groups_ms <= 0.0 || (double)ctx->listed_ms >= wave_groups_ms;
                    // ... and such a frame's direct kernel stays on the context's own stream, behind the listed kernel (a
                    // few ms on top of hundreds): sixteen contexts' streams and the shared ones are more streams than
                    // hardware queues, and a direct kernel that shares a queue with another frame's listed kernel waits
                    // for all of it (glass 1080p, 16 in flight: 22.4-22.8 -> 21.3-21.8 ms per frame)
                    static const int shared_n = std::min(RT_SEQ_DIRECT_MAX, getenv("RT_SEQ_DIRECT_STREAMS") ? atoi(getenv("RT_SEQ_DIRECT_STREAMS")) : RT_SEQ_DIRECT_STREAMS);
                    if (shared_n > 0 && s2 == ctx->stream && !no_alloc && ctx->counted && n_listed < npix && !tg.wave_groups) {
                        // this frame's direct kernel: onto the next shared stream, behind the classification (ev_list);
                        // the context's stream waits for it at the end of the frame (ev_gen).  One lock around the
                        // three calls, so that no other context's kernel gets between this kernel and its event.
                        // (no early return with the lock held: a failing call is reported after the lock is gone, and a
                        // stream's turn only advances once the stream exists)
                        hipError_t e = hipSuccess;
                        {
                            std::lock_guard<std::mutex> lock(g_direct_mutex);
                            SharedDirect& sd = g_direct[ctx->device];
                            hipStream_t& ds = sd.s[sd.turn % (unsigned)shared_n];
                            if (!ds) e = hipStreamCreateWithFlags(&ds, hipStreamNonBlocking);
                            if (e == hipSuccess) {
                                ++sd.turn;
                                ctx->shared_used = ds;
                                e = hipStreamWaitEvent(ds, ctx->ev_list, 0);
                                if (e == hipSuccess) e = launch_direct(a, t, ds);
                                if (e == hipSuccess) e = hipEventRecord(ctx->ev_gen, ds);
                            }
                        }
                        HIP_TRY(ctx, e);
                        direct_done = true; second = true;
                    }
                    // one launch owns the whole list (one dispatch less per frame; the split below buys latency for a
                    // frame on its own)
                    g2.listed_mode = RT_MODE_ALL;
                    HIP_TRY(ctx, hipEventRecord(ctx->ev_l0, s2));
                    HIP_TRY(ctx, rt_launch_render(g2, tg, s2));
                    HIP_TRY(ctx, hipEventRecord(ctx->ev_l1, s2));
                    ctx->listed_timed = true;
                    if (!no_handover) HIP_TRY(ctx, rt_launch_heavy(g2, tg, s2));
                } else if (g2.heavy_cap > 0) {
                    // the steady long pixels: first samples, hand-over, heavy kernel, resume kernel ...
                    g2.listed_mode = RT_MODE_STEADY;
                    HIP_TRY(ctx, rt_launch_render(g2, tg, s2));
                    HIP_TRY(ctx, rt_launch_heavy(g2, tg, s2));
                    // ... and beside them all other listed pixels
                    g2.listed_mode = RT_MODE_REST;
                    if (!ctx->stream3) HIP_TRY(ctx, hipStreamCreateWithFlags(&ctx->stream3, hipStreamNonBlocking));
                    HIP_TRY(ctx, hipStreamWaitEvent(ctx->stream3, ctx->ev_list, 0));
                    HIP_TRY(ctx, rt_launch_render(g2, tg, ctx->stream3));
                    HIP_TRY(ctx, hipEventRecord(ctx->ev_rest, ctx->stream3));
                    rest = true;
                } else {
                    HIP_TRY(ctx, rt_launch_render(g2, tg, s2));
                }
                if (s2 != ctx->stream) { HIP_TRY(ctx, hipEventRecord(ctx->ev_gen, s2)); second = true; }
            }
        }
        if (!timed) HIP_TRY(ctx, hipEventRecord(ctx->evm, ctx->stream));
        if (n_listed < npix && !direct_done) HIP_TRY(ctx, launch_direct(a, t, ctx->stream));
        if (second) HIP_TRY(ctx, hipStreamWaitEvent(ctx->stream, ctx->ev_gen, 0));
        if (rest) HIP_TRY(ctx, hipStreamWaitEvent(ctx->stream, ctx->ev_rest, 0));
    }
    HIP_TRY(ctx, hipEventRecord(ctx->ev1, ctx->stream));

    if (rgb_out && !rgb_dev) HIP_TRY(ctx, hipMemcpyAsync(rgb_out, a.rgb, npix * 3, hipMemcpyDeviceToHost, ctx->stream));
    if (emission_out && !em_dev) HIP_TRY(ctx, hipMemcpyAsync(emission_out, a.emission, npix * 16, hipMemcpyDeviceToHost, ctx->stream));
    unsigned long long words[6];
    HIP_TRY(ctx, hipMemcpyAsync(ctx->h_back, state, sizeof(words), hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, wait_stream(ctx, (p->flags & RT_FLAG_SEQUENCE) != 0));
    std::memcpy(words, ctx->h_back, sizeof(words));
    HIP_TRY(ctx, hipEventElapsedTime(&ctx->last_ms, ctx->evm, ctx->ev1));
    HIP_TRY(ctx, hipEventElapsedTime(&ctx->last_total_ms, ctx->ev0, ctx->ev1));

    int fault; std::memcpy(&fault, &words[5], sizeof(int));
    if (fault) return RT_ERR_DEVICE_FAULT;
    if (count) {
        counters->primary = (uint64_t)npix * (uint64_t)p->samples * 4u;
        counters->closest = words[0]; counters->shadow = words[1]; counters->tests = words[2]; counters->rand_draws = words[3]; counters->shaded = words[4];
    }
    return RT_OK;
}

extern "C" {

int rt_render_rows(rt_ctx* ctx, const rt_camera35f* cam, const rt_params* p,
                   uint8_t* rgb_out, float* emission_out, rt_counters* counters)
{
    return render_rows_impl(ctx, cam, p, rgb_out, emission_out, counters, false);
}

int rt_render_rows_static(rt_ctx* ctx, const rt_camera35f* cam, const rt_params* p, uint8_t* rgb_dev, float* emission_dev)
{
    return render_rows_impl(ctx, cam, p, rgb_dev, emission_dev, nullptr, true);
}

int rt_render_tiles(rt_ctx* ctx, const rt_camera35f* cam, const rt_params* p, int32_t tile, int32_t rows_per_chunk,
                    rt_tile_sink sink, void* user)
{
    if (!ctx || !cam || !p || !sink) return RT_ERR_INVALID_ARG;
    if (tile <= 0 || tile > 256 || rows_per_chunk < 0) return RT_ERR_INVALID_ARG;
    rt_params q = *p;
    q.row0 = 0; q.rows = p->height; q.band_rows = 0; q.band_stride = 0;
    q.flags &= ~(RT_FLAG_COUNT_LIKE_REFERENCE | RT_FLAG_COUNT_EXECUTED);
    int rc = check_params(ctx, cam, &q);
    if (rc != RT_OK) return rc;
    const int tiles_x = p->width / tile, tiles_y = p->height / tile;          // src/main.cpp:276: whole tiles only
    if (tiles_x == 0 || tiles_y == 0) return RT_OK;
    int chunk_tiles_y = rows_per_chunk == 0 ? tiles_y : rows_per_chunk / tile;
    if (chunk_tiles_y < 1) chunk_tiles_y = 1;
    if (chunk_tiles_y > tiles_y) chunk_tiles_y = tiles_y;

    DeviceGuard g(ctx->device);
    if (!g.ok) return RT_ERR_NO_DEVICE;
    const size_t img_bytes = 8 + (size_t)tile * tile * 3, em_bytes = 8 + (size_t)tile * tile * 16;
    const size_t chunk_rows = (size_t)chunk_tiles_y * tile, chunk_tiles = (size_t)chunk_tiles_y * tiles_x;
    const size_t rgb_bytes = (chunk_rows * p->width * 3 + 15) & ~(size_t)15, emi_bytes = chunk_rows * p->width * 16;
    const size_t irec_bytes = (chunk_tiles * img_bytes + 15) & ~(size_t)15, erec_bytes = chunk_tiles * em_bytes;
    rc = grow(ctx, &ctx->d_tiles, &ctx->tiles_bytes, rgb_bytes + emi_bytes + irec_bytes + erec_bytes);
    if (rc) return rc;
    unsigned char* d_rgb = (unsigned char*)ctx->d_tiles;
    unsigned char* d_emi = d_rgb + rgb_bytes;
    unsigned char* d_irec = d_emi + emi_bytes;
    unsigned char* d_erec = d_irec + irec_bytes;
    std::vector<unsigned char> h_rec(irec_bytes + erec_bytes);

    for (int ty0 = 0; ty0 < tiles_y; ty0 += chunk_tiles_y) {
        const int nty = ty0 + chunk_tiles_y <= tiles_y ? chunk_tiles_y : tiles_y - ty0;
        q.row0 = ty0 * tile; q.rows = nty * tile;
        rc = rt_render_rows(ctx, cam, &q, d_rgb, (float*)d_emi, nullptr);
        if (rc != RT_OK) return rc;
        const int n = nty * tiles_x;
        HIP_TRY(ctx, rt_launch_tile_pack(d_rgb, d_emi, p->width, tile, tiles_x, ty0, n, d_irec, d_erec, ctx->stream));
        HIP_TRY(ctx, hipMemcpyAsync(h_rec.data(), d_irec, (size_t)n * img_bytes, hipMemcpyDeviceToHost, ctx->stream));
        HIP_TRY(ctx, hipMemcpyAsync(h_rec.data() + irec_bytes, d_erec, (size_t)n * em_bytes, hipMemcpyDeviceToHost, ctx->stream));
        HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
        for (int t = 0; t < n; ++t) {                                         // src/main.cpp:325-326: image, then emission
            if (sink(user, h_rec.data() + (size_t)t * img_bytes, img_bytes) != 0) return RT_ERR_INVALID_ARG;
            if (sink(user, h_rec.data() + irec_bytes + (size_t)t * em_bytes, em_bytes) != 0) return RT_ERR_INVALID_ARG;
        }
    }
    return RT_OK;
}

int rt_probe_max_depth(rt_ctx* ctx, const rt_camera35f* cam, const rt_params* p, uint8_t* flags_out)
{
    int rc = check_params(ctx, cam, p);
    if (rc != RT_OK) return rc;
    if (!flags_out) return RT_ERR_INVALID_ARG;
    DeviceGuard g(ctx->device);
    if (!g.ok) return RT_ERR_NO_DEVICE;

    rc = refresh_primary_terms(ctx, cam);
    if (rc != RT_OK) return rc;
    KernelArgs a;
    fill_args(ctx, cam, p, a);
    a.row0 = 0; a.rows = p->height; a.band_rows = 0; a.band_stride = 0;
    a.tiles_y = (p->height + 15) / 16;
    const size_t npix = (size_t)p->width * p->height;
    const bool dev = is_device_pointer(flags_out);
    if (dev) a.rgb = flags_out;
    else { rc = grow(ctx, &ctx->d_stage_rgb, &ctx->stage_rgb_bytes, npix); if (rc) return rc; a.rgb = (unsigned char*)ctx->d_stage_rgb; }
    LaunchTraits t = { false, true, true, true, false, false, true };
    HIP_TRY(ctx, hipMemsetAsync(ctx->d_counters, 0, RT_DEV_WORDS_BYTES, ctx->stream));
    HIP_TRY(ctx, rt_launch_probe(a, t, ctx->stream));
    if (!dev) HIP_TRY(ctx, hipMemcpyAsync(flags_out, a.rgb, npix, hipMemcpyDeviceToHost, ctx->stream));
    int fault = 0;
    HIP_TRY(ctx, hipMemcpyAsync(&fault, ctx->d_fault, sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return fault ? RT_ERR_DEVICE_FAULT : RT_OK;
}

// every row of every band must exist in its rank's block of rows_per_rank rows: band b is rank b % nranks' band
// b / nranks (the last band of the frame may be short, an earlier full band of another rank then needs the larger index)
static bool blocks_hold_frame(int height, int band_rows, int nranks, int rows_per_rank)
{
    const int nbands = (height + band_rows - 1) / band_rows;
    for (int b = nbands - 1; b >= 0 && b >= nbands - nranks; --b) {          // the last band of each rank
        const int rows = (b == nbands - 1) ? height - b * band_rows : band_rows;
        if ((b / nranks) * band_rows + rows > rows_per_rank) return false;
    }
    return true;
}

int rt_deinterleave_bands(rt_ctx* ctx, const void* gathered, void* frame_out, int32_t width, int32_t height,
                          int32_t bytes_per_pixel, int32_t band_rows, int32_t nranks, int32_t rows_per_rank)
{
    if (!ctx || !gathered || !frame_out) return RT_ERR_INVALID_ARG;
    if (width <= 0 || height <= 0 || bytes_per_pixel <= 0 || band_rows <= 0 || nranks <= 0 || rows_per_rank <= 0) return RT_ERR_INVALID_ARG;
    if (!blocks_hold_frame(height, band_rows, nranks, rows_per_rank)) return RT_ERR_INVALID_ARG;
    if (!is_device_pointer(gathered) || !is_device_pointer(frame_out)) return RT_ERR_INVALID_ARG;
    DeviceGuard g(ctx->device);
    if (!g.ok) return RT_ERR_NO_DEVICE;
    HIP_TRY(ctx, rt_launch_deinterleave(gathered, frame_out, width * bytes_per_pixel, height, band_rows, nranks, rows_per_rank, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return RT_OK;
}

// ---- the band gather over RCCL (include/rt_amd.h "multi-GPU") --------------------------------------------------
// RCCL is bound at run time, on the first call that needs it: a one-GPU caller never loads it, and a process that has
// loaded one already (torch brings its own) keeps using that one.
}  // extern "C"

namespace {
struct Rccl {
    void* lib = nullptr;
    decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
    decltype(&ncclCommInitRank) CommInitRank = nullptr;
    decltype(&ncclCommDestroy) CommDestroy = nullptr;
    decltype(&ncclCommCount) CommCount = nullptr;
    decltype(&ncclGroupStart) GroupStart = nullptr;
    decltype(&ncclGroupEnd) GroupEnd = nullptr;
    decltype(&ncclSend) Send = nullptr;
    decltype(&ncclRecv) Recv = nullptr;
    bool ok = false;
};
Rccl g_rccl;
std::once_flag g_rccl_once;

const Rccl& rccl()
{
    std::call_once(g_rccl_once, [] {
        Rccl& r = g_rccl;
        for (const char* name : { "librccl.so.1", "librccl.so" }) {
            r.lib = dlopen(name, RTLD_NOW | RTLD_NOLOAD | RTLD_GLOBAL);      // one that the process has already
            if (r.lib) break;
        }
        for (const char* name : { "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1" }) {
            if (r.lib) break;
            r.lib = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
        }
        if (!r.lib) return;
        r.GetUniqueId = (decltype(r.GetUniqueId))dlsym(r.lib, "ncclGetUniqueId");
        r.CommInitRank = (decltype(r.CommInitRank))dlsym(r.lib, "ncclCommInitRank");
        r.CommDestroy = (decltype(r.CommDestroy))dlsym(r.lib, "ncclCommDestroy");
        r.CommCount = (decltype(r.CommCount))dlsym(r.lib, "ncclCommCount");
        r.GroupStart = (decltype(r.GroupStart))dlsym(r.lib, "ncclGroupStart");
        r.GroupEnd = (decltype(r.GroupEnd))dlsym(r.lib, "ncclGroupEnd");
        r.Send = (decltype(r.Send))dlsym(r.lib, "ncclSend");
        r.Recv = (decltype(r.Recv))dlsym(r.lib, "ncclRecv");
        r.ok = r.GetUniqueId && r.CommInitRank && r.CommDestroy && r.GroupStart && r.GroupEnd && r.Send && r.Recv;
    });
    return g_rccl;
}
}  // namespace

extern "C" {

#define RCCL_TRY(ctx, expr)                                                                   \
    do {                                                                                      \
        ncclResult_t r__ = (expr);                                                            \
        if (r__ != ncclSuccess) { if (ctx) (ctx)->last_rccl = (int)r__; return RT_ERR_RCCL; } \
    } while (0)

static_assert(RT_COMM_ID_BYTES == NCCL_UNIQUE_ID_BYTES, "rt_comm_unique_id hands out an ncclUniqueId");

int rt_comm_unique_id(unsigned char* id)
{
    if (!id) return RT_ERR_INVALID_ARG;
    const Rccl& r = rccl();
    if (!r.ok) return RT_ERR_RCCL;
    ncclUniqueId u;
    if (r.GetUniqueId(&u) != ncclSuccess) return RT_ERR_RCCL;
    std::memcpy(id, u.internal, RT_COMM_ID_BYTES);
    return RT_OK;
}

int rt_comm_create(void** comm_out, rt_ctx* ctx, const unsigned char* id, int32_t rank, int32_t nranks)
{
    if (!comm_out) return RT_ERR_INVALID_ARG;
    *comm_out = nullptr;
    if (!ctx || !id || nranks <= 0 || rank < 0 || rank >= nranks) return RT_ERR_INVALID_ARG;
    const Rccl& r = rccl();
    if (!r.ok) return RT_ERR_RCCL;
    DeviceGuard g(ctx->device);                       // the communicator belongs to the context's GPU
    if (!g.ok) return RT_ERR_NO_DEVICE;
    ncclUniqueId u;
    std::memcpy(u.internal, id, RT_COMM_ID_BYTES);
    ncclComm_t comm = nullptr;
    RCCL_TRY(ctx, r.CommInitRank(&comm, nranks, u, rank));
    *comm_out = (void*)comm;
    return RT_OK;
}

void rt_comm_destroy(void* comm)
{
    if (comm && rccl().ok) (void)rccl().CommDestroy((ncclComm_t)comm);
}

int rt_comm_count(void* comm, int32_t* nranks_out)
{
    if (!comm || !nranks_out) return RT_ERR_INVALID_ARG;
    const Rccl& r = rccl();
    if (!r.ok || !r.CommCount) return RT_ERR_RCCL;
    int n = 0;
    if (r.CommCount((ncclComm_t)comm, &n) != ncclSuccess) return RT_ERR_RCCL;
    *nranks_out = n;
    return RT_OK;
}

int rt_last_rccl_error(const rt_ctx* ctx) { return ctx ? ctx->last_rccl : 0; }

int rt_gather_rows(rt_ctx* ctx, void* nccl_comm, int32_t rank, int32_t nranks, const void* my_block, void* gathered_on_rank0,
                   void* frame_on_rank0, int32_t width, int32_t height, int32_t bytes_per_pixel, int32_t band_rows, int32_t rows_per_rank)
{
    if (!ctx || !my_block) return RT_ERR_INVALID_ARG;
    if (width <= 0 || height <= 0 || bytes_per_pixel <= 0 || band_rows <= 0 || nranks <= 0 || rows_per_rank <= 0) return RT_ERR_INVALID_ARG;
    if (rank < 0 || rank >= nranks) return RT_ERR_INVALID_ARG;
    if (nranks > 1 && !nccl_comm) return RT_ERR_INVALID_ARG;
    if (rank == 0 && nranks > 1 && !gathered_on_rank0) return RT_ERR_INVALID_ARG;
    if (!blocks_hold_frame(height, band_rows, nranks, rows_per_rank)) return RT_ERR_INVALID_ARG;
    if (!is_device_pointer(my_block)) return RT_ERR_INVALID_ARG;
    if (rank == 0 && ((gathered_on_rank0 && !is_device_pointer(gathered_on_rank0)) || (frame_on_rank0 && !is_device_pointer(frame_on_rank0))))
        return RT_ERR_INVALID_ARG;
    DeviceGuard g(ctx->device);
    if (!g.ok) return RT_ERR_NO_DEVICE;
    const size_t block_bytes = (size_t)rows_per_rank * width * bytes_per_pixel;
    const void* blocks = my_block;                    // one rank without a communicator: its block is the gather buffer
    HIP_TRY(ctx, hipEventRecord(ctx->ev_g0, ctx->stream));
    if (nccl_comm) {
        // every rank sends its block to rank 0, which receives one block per rank (its own too): one group, so the
        // transfers of all peers run side by side, one xGMI link each
        const Rccl& r = rccl();
        if (!r.ok) return RT_ERR_RCCL;
        ncclComm_t comm = (ncclComm_t)nccl_comm;
        if (rank == 0 && !gathered_on_rank0) return RT_ERR_INVALID_ARG;
        RCCL_TRY(ctx, r.GroupStart());
        ncclResult_t e = r.Send(my_block, block_bytes, ncclUint8, 0, comm, ctx->stream);
        for (int peer = 0; rank == 0 && e == ncclSuccess && peer < nranks; ++peer)
            e = r.Recv((unsigned char*)gathered_on_rank0 + (size_t)peer * block_bytes, block_bytes, ncclUint8, peer, comm, ctx->stream);
        const ncclResult_t e2 = r.GroupEnd();
        RCCL_TRY(ctx, e);
        RCCL_TRY(ctx, e2);
        blocks = gathered_on_rank0;
    }
    if (rank == 0 && frame_on_rank0)
        HIP_TRY(ctx, rt_launch_deinterleave(blocks, frame_on_rank0, width * bytes_per_pixel, height, band_rows, nranks, rows_per_rank, ctx->stream));
    HIP_TRY(ctx, hipEventRecord(ctx->ev_g1, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));  // the block may be rendered into again, the frame read
    HIP_TRY(ctx, hipEventElapsedTime(&ctx->last_gather_ms, ctx->ev_g0, ctx->ev_g1));
    return RT_OK;
}

int rt_render_gather(rt_ctx* ctx, const rt_camera35f* cam, const rt_params* params, void* nccl_comm, int32_t rank, int32_t nranks,
                     int32_t band_rows, uint8_t* rgb_out_rank0)
{
    if (!ctx || !cam || !params || band_rows <= 0 || nranks <= 0 || rank < 0 || rank >= nranks) return RT_ERR_INVALID_ARG;
    if (nranks > 1 && !nccl_comm) return RT_ERR_INVALID_ARG;
    if (rank == 0 && !rgb_out_rank0) return RT_ERR_INVALID_ARG;
    if (params->width <= 0 || params->height <= 0) return RT_ERR_INVALID_ARG;
    // this rank's cyclic bands of the whole frame (band b -> rank b % nranks), packed into a block of the size every rank uses
    const int W = params->width, H = params->height;
    const int nbands = (H + band_rows - 1) / band_rows;
    const int rows_per_rank = ((nbands + nranks - 1) / nranks) * band_rows;
    int my_rows = 0;
    for (int b = rank; b < nbands; b += nranks) my_rows += (b == nbands - 1) ? H - b * band_rows : band_rows;
    rt_params p = *params;
    p.row0 = rank * band_rows; p.rows = my_rows; p.band_rows = band_rows; p.band_stride = band_rows * nranks;
    p.flags &= ~(RT_FLAG_COUNT_LIKE_REFERENCE | RT_FLAG_COUNT_EXECUTED);
    const size_t block_bytes = (size_t)rows_per_rank * W * 3, frame_bytes = (size_t)H * W * 3;
    const bool out_dev = rank == 0 && is_device_pointer(rgb_out_rank0);
    // block | (rank 0) gather buffer | (rank 0, host output) frame
    const size_t need = block_bytes + (rank == 0 ? (size_t)nranks * block_bytes + (out_dev ? 0 : frame_bytes) : 0);
    int rc = grow(ctx, &ctx->d_gather, &ctx->gather_bytes, need);
    if (rc) return rc;
    unsigned char* block = (unsigned char*)ctx->d_gather;
    unsigned char* gathered = block + block_bytes;
    unsigned char* frame = out_dev ? rgb_out_rank0 : gathered + (size_t)nranks * block_bytes;
    if (my_rows > 0) {
        rc = rt_render_rows(ctx, cam, &p, block, nullptr, nullptr);
        if (rc != RT_OK) return rc;
    }
    rc = rt_gather_rows(ctx, nccl_comm, rank, nranks, block, rank == 0 ? gathered : nullptr, rank == 0 ? frame : nullptr, W, H, 3,
                        band_rows, rows_per_rank);
    if (rc != RT_OK) return rc;
    if (rank == 0 && !out_dev) {
        DeviceGuard g(ctx->device);
        if (!g.ok) return RT_ERR_NO_DEVICE;
        HIP_TRY(ctx, hipMemcpyAsync(rgb_out_rank0, frame, frame_bytes, hipMemcpyDeviceToHost, ctx->stream));
        HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    }
    return RT_OK;
}

// applyGlow (src/main.cpp:163-229) on the GPU: rt_glow.hip has the kernels.  The loop over pyramid levels runs here, as
// in the reference; each level's largest strength comes back to decide whether another level follows (src/main.cpp:173).
int rt_apply_glow(rt_ctx* ctx, uint8_t* rgb, const float* emission, int32_t width, int32_t height, int32_t* levels_out)
{
    if (levels_out) *levels_out = 0;
    if (!ctx || !rgb || !emission || width <= 0 || height <= 0) return RT_ERR_INVALID_ARG;
    if ((int64_t)width * height > (int64_t)1 << 30) return RT_ERR_INVALID_ARG;
    DeviceGuard g(ctx->device);
    if (!g.ok) return RT_ERR_NO_DEVICE;
    const size_t npix = (size_t)width * height;
    const bool rgb_dev = is_device_pointer(rgb), em_dev = is_device_pointer(emission);
    // two level buffers (a level is at most a quarter of the frame; the first input is the frame's emission itself) |
    // the blur's weights | the frame and its emission when they come from the host | the level's largest strength
    const size_t level_bytes = ((npix / 4 + 1) * 16 + 255) & ~(size_t)255;
    const size_t max_kernel = 257 * 257;                        // weights are uploaded per level; larger windows: see below
    const size_t kernel_bytes = (max_kernel * 4 + 255) & ~(size_t)255;
    size_t need = 2 * level_bytes + kernel_bytes + 256;
    const size_t rgb_at = need; if (!rgb_dev) need += (npix * 3 + 255) & ~(size_t)255;
    const size_t em_at = need;  if (!em_dev) need += npix * 16;
    int rc = grow(ctx, &ctx->d_glow, &ctx->glow_bytes, need);
    if (rc) return rc;
    unsigned char* base = (unsigned char*)ctx->d_glow;
    void* level[2] = { base, base + level_bytes };
    float* d_kernel = (float*)(base + 2 * level_bytes);
    unsigned int* d_max = (unsigned int*)(base + 2 * level_bytes + kernel_bytes);
    unsigned char* d_rgb = rgb_dev ? rgb : base + rgb_at;
    const void* d_em = em_dev ? (const void*)emission : (const void*)(base + em_at);
    if (!rgb_dev) HIP_TRY(ctx, hipMemcpyAsync(d_rgb, rgb, npix * 3, hipMemcpyHostToDevice, ctx->stream));
    if (!em_dev) HIP_TRY(ctx, hipMemcpyAsync((void*)d_em, emission, npix * 16, hipMemcpyHostToDevice, ctx->stream));

    float max = 1.0f;
    const int downScaleFactor = 2;
    int upScaleFactor = downScaleFactor;
    const int kernelSigma = (int)1000.0f;                       // `int kernelSigma = 1000.0f` (src/main.cpp:168)
    int kernelSize = 8;
    int w = width, h = height, levels = 0;
    const void* src = d_em;
    int status = RT_OK;
    while (max >= 1 && w > 0 && h > 0) {
        const int dW = w / downScaleFactor, dH = h / downScaleFactor;
        // the reference reads upscaled[x + y*startW] from rows that are dW*upScaleFactor long: out of bounds unless equal
        if (dW * upScaleFactor != width || dH * upScaleFactor != height) { status = RT_ERR_UNSUPPORTED; break; }
        const size_t n = (size_t)(2 * kernelSize + 1);
        if (n * n > max_kernel) { status = RT_ERR_UNSUPPORTED; break; }                  // (a 5th level of a frame > 4096 wide)
        // the blur's weights (src/main.cpp:101-123): exp() in double, sums in float, in the reference's order
        std::vector<float>& kernel = ctx->glow_kernels[levels < 8 ? levels : 7];       // kept: the same for every frame
        if (kernel.size() != n * n) {
            kernel.assign(n * n, 0.0f);
            const float sigma = (float)kernelSigma;
            float sum = 0.0f;
            for (int y = -kernelSize; y <= kernelSize; ++y)
                for (int x = -kernelSize; x <= kernelSize; ++x) {
                    const float value = (float)std::exp((double)(-(x * x + y * y) / (2 * sigma * sigma)));
                    kernel[(size_t)(x + kernelSize) * n + (size_t)(y + kernelSize)] = value;
                    sum += value;
                }
            sum = 1.0f / sum;
            for (size_t y = 0; y < n; ++y)
                for (size_t x = 0; x < n; ++x) kernel[x * n + y] *= sum;
        }
        HIP_TRY(ctx, hipMemcpyAsync(d_kernel, kernel.data(), n * n * 4, hipMemcpyHostToDevice, ctx->stream));
        HIP_TRY(ctx, hipMemsetAsync(d_max, 0, 4, ctx->stream));
        // down: level[0], blurred: level[1].  From the second level on the source is the previous level's blurred image in
        // level[1]: the downsample has read it before the blur (next on the stream) writes there again
        void* down = level[0];
        void* blurred = level[1];
        HIP_TRY(ctx, rt_launch_glow_downsample(src, w, down, dW, dH, ctx->stream));
        HIP_TRY(ctx, rt_launch_glow_blur(down, blurred, dW, dH, kernelSize, d_kernel, ctx->stream));
        HIP_TRY(ctx, rt_launch_glow_add(d_rgb, blurred, dW, upScaleFactor, width, height, ctx->stream));
        HIP_TRY(ctx, rt_launch_glow_filter(blurred, dW * dH, d_max, ctx->stream));
        unsigned int bits = 0;
        HIP_TRY(ctx, hipMemcpyAsync(&bits, d_max, 4, hipMemcpyDeviceToHost, ctx->stream));
        HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));        // (also: `kernel` may be overwritten)
        std::memcpy(&max, &bits, 4);
        w = dW; h = dH;
        src = blurred;
        upScaleFactor *= 2;
        kernelSize *= 2;
        ++levels;
    }
    if (!rgb_dev) HIP_TRY(ctx, hipMemcpyAsync(rgb, d_rgb, npix * 3, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    if (levels_out) *levels_out = levels;
    return status;
}

int rt_debug_rand(rt_ctx* ctx, const uint32_t* seeds, int32_t n_seeds, int32_t n_draws, int32_t* out)
{
    if (!ctx || !seeds || !out || n_seeds <= 0 || n_draws <= 0) return RT_ERR_INVALID_ARG;
    DeviceGuard g(ctx->device);
    if (!g.ok) return RT_ERR_NO_DEVICE;
    unsigned int* d_seeds = nullptr; int* d_out = nullptr;
    HIP_TRY(ctx, hipMalloc((void**)&d_seeds, sizeof(uint32_t) * n_seeds));
    hipError_t e = hipMalloc((void**)&d_out, sizeof(int32_t) * (size_t)n_seeds * n_draws);
    if (e == hipSuccess) e = hipMemcpyAsync(d_seeds, seeds, sizeof(uint32_t) * n_seeds, hipMemcpyHostToDevice, ctx->stream);
    if (e == hipSuccess) e = rt_launch_debug_rand(d_seeds, n_seeds, n_draws, d_out, ctx->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(out, d_out, sizeof(int32_t) * (size_t)n_seeds * n_draws, hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    (void)hipFree(d_seeds); if (d_out) (void)hipFree(d_out);
    HIP_TRY(ctx, e);
    return RT_OK;
}

int rt_debug_exact_math(rt_ctx* ctx, uint64_t* mismatches3)
{
    if (!ctx || !mismatches3) return RT_ERR_INVALID_ARG;
    DeviceGuard g(ctx->device);
    if (!g.ok) return RT_ERR_NO_DEVICE;
    unsigned long long* d_out = nullptr;
    HIP_TRY(ctx, hipMalloc((void**)&d_out, 3 * sizeof(unsigned long long)));
    hipError_t e = hipMemsetAsync(d_out, 0, 3 * sizeof(unsigned long long), ctx->stream);
    if (e == hipSuccess) e = rt_launch_debug_exact_math(d_out, ctx->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(mismatches3, d_out, 3 * sizeof(unsigned long long), hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    (void)hipFree(d_out);
    HIP_TRY(ctx, e);
    return RT_OK;
}

int rt_debug_quantise(rt_ctx* ctx, const float* in, int32_t n, uint8_t* out)
{
    if (!ctx || !in || !out || n <= 0) return RT_ERR_INVALID_ARG;
    DeviceGuard g(ctx->device);
    if (!g.ok) return RT_ERR_NO_DEVICE;
    float* d_in = nullptr; unsigned char* d_out = nullptr;
    HIP_TRY(ctx, hipMalloc((void**)&d_in, sizeof(float) * (size_t)n));
    hipError_t e = hipMalloc((void**)&d_out, (size_t)n);
    if (e == hipSuccess) e = hipMemcpyAsync(d_in, in, sizeof(float) * (size_t)n, hipMemcpyHostToDevice, ctx->stream);
    if (e == hipSuccess) e = rt_launch_debug_quantise(d_in, n, d_out, ctx->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(out, d_out, (size_t)n, hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    (void)hipFree(d_in); if (d_out) (void)hipFree(d_out);
    HIP_TRY(ctx, e);
    return RT_OK;
}

int rt_debug_intersect(rt_ctx* ctx, int32_t index, const float* rays, int32_t n_rays, float* out8)
{
    if (!ctx || !rays || !out8 || n_rays <= 0) return RT_ERR_INVALID_ARG;
    if (!ctx->has_scene) return RT_ERR_NO_SCENE;
    if (index < 0 || index >= ctx->n_hit) return RT_ERR_INVALID_ARG;
    DeviceGuard g(ctx->device);
    if (!g.ok) return RT_ERR_NO_DEVICE;
    float* d_rays = nullptr; float* d_out = nullptr;
    HIP_TRY(ctx, hipMalloc((void**)&d_rays, sizeof(float) * 6 * (size_t)n_rays));
    hipError_t e = hipMalloc((void**)&d_out, sizeof(float) * 8 * (size_t)n_rays);
    if (e == hipSuccess) e = hipMemcpyAsync(d_rays, rays, sizeof(float) * 6 * (size_t)n_rays, hipMemcpyHostToDevice, ctx->stream);
    if (e == hipSuccess) e = rt_launch_debug_intersect(ctx->d_hrec, index, d_rays, n_rays, d_out, ctx->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(out8, d_out, sizeof(float) * 8 * (size_t)n_rays, hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    (void)hipFree(d_rays); if (d_out) (void)hipFree(d_out);
    HIP_TRY(ctx, e);
    return RT_OK;
}

}  // extern "C"
