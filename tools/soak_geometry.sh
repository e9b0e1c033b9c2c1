#!/bin/bash
# One-off soak of the run-time-geometry and full-band featuriser fuzz tests over several seeds, in ONE process (GPU box).
# Prints as it goes (no pipe into grep / tail: a silent run is taken to be hung after 7 minutes).
cd tests && timeout -k 10 900 python -c "
import os, sys, warnings
warnings.simplefilter('ignore')
sys.path.insert(0, '.'); sys.path.insert(0, '..')
import test_gpu_fuzz as t
for s in (606, 11, 12, 13):
    os.environ['COUGH_FUZZ_SEED_GEO'] = str(s)
    try:
        t.test_runtime_geometry_featuriser_random_stft_geometries()
        print('geometry seed', s, 'ok', flush=True)
    except AssertionError as e:
        print('geometry seed', s, 'FAILED', str(e)[:600], flush=True)
for s in (505, 21):
    os.environ['COUGH_FUZZ_SEED_FULLBAND'] = str(s)
    try:
        t.test_fullband_featuriser_random_filterbanks_and_flags()
        print('full-band seed', s, 'ok', flush=True)
    except AssertionError as e:
        print('full-band seed', s, 'FAILED', str(e)[:600], flush=True)
" 2>&1
