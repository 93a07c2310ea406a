#!/bin/bash
# A/B/... of builds of the library on ONE box (clocks differ by ~1 % between boxes): alternating default bench runs.
# usage (GPU box, repo root): bash tools/ab_bench.sh cora_amd/libcorahip_base.so cora_amd/libcorahip.so [more libs] [-- bench args]
LIBS=(); while [ $# -gt 0 ] && [ "$1" != "--" ]; do LIBS+=("$1"); shift; done; [ "$1" == "--" ] && shift
for rep in 1 2 3; do
  for L in "${LIBS[@]}"; do
    CORAHIP_LIB=$PWD/$L python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-host-delivered --no-seeded-modes "$@" 2>/dev/null | \
      python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('$L', round(d['ms_per_step'],2), d['stages_ms'], round(d['roofline']['frac'],4))" || exit 1
  done
done
