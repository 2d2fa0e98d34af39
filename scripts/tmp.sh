for l in 1 0 1; do echo "lds=$l"; AH_ROWMAJOR_LDS=$l AH_TIMING=1 python bench.py --steps 2 --warmup 1 --no-cpu 2>&1 | grep -E "^\[ah\]"; done
