#!/bin/bash
# Same-box A/B of several builds of libcough_amd.so: interleaved bench runs.
# Usage: bash tools/ab.sh <libA> <libB> [<libC> ...] [-- bench args]
LIBS=()
while [ $# -gt 0 ] && [ "$1" != "--" ]; do LIBS+=("$1"); shift; done
[ "$1" == "--" ] && shift
for i in 1 2 3; do
  for L in "${LIBS[@]}"; do
    COUGH_AMD_LIB=$L timeout -k 10 100 python bench.py --steps 60 --warmup 10 --cpu-seconds 0 "$@" 2>/dev/null | python3 -c "
import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; c=d.get('roofline_classifier',{})
print('$L'.split('/')[-1], d['value'], d['ms_per_step'], 'k1', r['ms_per_launch'], 'cls', c.get('ms_per_forward'), 'stft', d.get('roofline_stft', {}).get('ms_per_launch'))"
  done
done
