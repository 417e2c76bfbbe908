for n in 2 4 8 16; do for f in 0 8; do echo "ranks $n flags $f"; RT_EMULATE_RANKS=$n timeout -k 10 100 python tools/time_workload.py default4_4k_64spp $f 2>/dev/null | tail -2; done; done
