#!/bin/bash
# round 5: randomised GPU-vs-oracle sweeps on the round's library (block descent, narrow dense kernel, guarded entry points)
OUT=gpurun_out/r05fuzz
mkdir -p $OUT
cd "$GRAFT_REPO_ROOT"
{
echo "# scripts/fuzz_gpu.py on libarroy_hip.so sha256 $(sha256sum arroy_amd/libarroy_hip.so | cut -c1-16)...: random shapes, metrics, scales, id layouts,"
echo "# margin modes (narrow dense kernel at 64 / 128 / 256 columns or off), streamed builds, filters, every search path (block / wave / octet"
echo "# descent x tiles / sorted / unscreened re-rank) — each configuration compared with the CPU oracle bit for bit."
for cfg in "150 71" "150 72 AH_SCREEN_VERIFY=1" "150 73" "120 74 AH_SCREEN_VERIFY=1"; do
  set -- $cfg
  secs=$1; seed=$2; shift 2
  printf "%-70s" "$* python scripts/fuzz_gpu.py $secs $seed"
  env "$@" timeout 600 python scripts/fuzz_gpu.py $secs $seed 2>&1 | tail -1
done
} | tee $OUT/r05_fuzz.txt
