#!/bin/bash
# tools/gpu_call.sh TAG TIMEOUT_S 'command ...' - one gpurun call whose record stays beside its logs.
# The command runs on the GPU box from the repo root and should write its logs under gpurun_out/TAG/; when the call
# ends, gpurun's own verdict of it (gpurun_out/.last_call.json: exit code, gpu_fault, stderr tail - overwritten by the
# next call) is copied to gpurun_out/TAG/call.json, so a crash can still be read after later calls.
# GPU-box pytest runs go through pytest.ini (--capture=sys: fd 2 stays the log) and tests/conftest.py (abort_trace).
set -u
tag=$1; secs=$2; shift 2
mkdir -p "$(dirname "$0")/../gpurun_out/$tag"
/usr/local/graft/bin/gpurun --timeout "$secs" -- "mkdir -p gpurun_out/$tag && ( $* )"
rc=$?
root="$(dirname "$0")/.."
[ -f "$root/gpurun_out/.last_call.json" ] && cp "$root/gpurun_out/.last_call.json" "$root/gpurun_out/$tag/call.json"
exit $rc
