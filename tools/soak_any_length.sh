#!/bin/bash
# One-off soak of tests/test_gpu_fuzz.py::test_random_waveform_lengths_through_random_handles over several seeds, in ONE process.
# Usage on the GPU box: bash tools/soak_any_length.sh   (8 seeds x 24 configurations x 3 lengths: all passed, round 5)
export COUGH_FUZZ_CASES=24
cd tests && timeout -k 10 600 python -c "
import os, sys
sys.path.insert(0, '.'); sys.path.insert(0, '..')
import test_gpu_fuzz as t
for s in (707, 1, 2, 3, 4, 5, 6, 7):
    os.environ['COUGH_FUZZ_SEED'] = str(s)
    t.test_random_waveform_lengths_through_random_handles()
    print('seed', s, 'ok', flush=True)
" 2>&1 | grep -v Warning | tail -12
