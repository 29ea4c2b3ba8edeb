#!/bin/bash
# Runs the GPU test-suite once per setting of RT_CLIP given on the command line (default: the library's default).
# usage (GPU box, repository root): tools/gpu_tests.sh [0|1 ...]
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out
for clip in "${@:-default}"; do
  if [ "$clip" = default ]; then unset RT_CLIP; else export RT_CLIP=$clip; fi
  echo "== RT_CLIP=$clip"
  timeout -k 10 1000 python -u -m pytest tests -m gpu -x -q --timeout 300 --timeout-method thread > gpurun_out/gputests_clip_$clip.log 2>&1
  rc=$?
  tail -15 gpurun_out/gputests_clip_$clip.log
  if [ $rc -ne 0 ]; then echo "FAILED rc=$rc"; exit $rc; fi
done
