#!/bin/bash
# The whole `-m gpu` suite against the poison build (make -C sigmarl_amd/csrc poison), once per round; and -- when the two `_revert` libraries exist (the tree with the
# fix of ea57876 taken out again, built by hand) -- how long the poison build needs to catch the round-4 bug.  Output: gpurun_out/<tag>/poison_suite.txt
tag=${1:-r06_poison}; out=$GRAFT_REPO_ROOT/gpurun_out/$tag; mkdir -p "$out"
C=$GRAFT_REPO_ROOT/sigmarl_amd/csrc
{
echo "== python -m pytest tests -m gpu against libsigmaenv_poison.so"
( time SIGMAENV_LIB=$C/libsigmaenv_poison.so python -m pytest tests -m gpu -q 2>&1 | tail -6 ) 2>&1
for lib in libsigmaenv_poison_revert.so libsigmaenv_revert.so; do
  [ -f $C/$lib ] || continue
  echo; echo "== the packed Hessian's last word NOT cleared (ea57876 taken out): $lib, three runs of the regression test + the 17..64-vehicle fuzz test"
  for r in 1 2 3; do ( time SIGMAENV_ALLOW_STALE=1 SIGMAENV_LIB=$C/$lib python -m pytest tests/test_gpu_cbf.py -q -k "odd_vehicle_count" 2>&1 | tail -2 ) 2>&1 | grep -E "passed|failed|real"; done
  ( time SIGMAENV_ALLOW_STALE=1 SIGMAENV_LIB=$C/$lib python -m pytest tests/test_gpu_fuzz.py -q -k "17_to_64" 2>&1 | tail -2 ) 2>&1 | grep -E "passed|failed|real"
done
} | tee "$out/poison_suite.txt"
