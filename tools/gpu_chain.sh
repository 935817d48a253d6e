#!/bin/bash
# Runs GPU steps one after another on the dev box, each under its own `timeout -k`, and stops the chain as soon as one was KILLED at
# its limit (a hung GPU step must not be followed by another); an ordinary failure (a red test) lets the later steps run.
#   tools/gpu_chain.sh "<secs> <log> <command...>" ...
mkdir -p gpurun_out
for step in "$@"; do
    secs=${step%% *}; rest=${step#* }; log=${rest%% *}; cmd=${rest#* }
    echo "[chain] ${cmd}  (limit ${secs}s) -> gpurun_out/${log}"
    timeout -k 10 "${secs}" bash -c "${cmd}" > "gpurun_out/${log}" 2>&1
    rc=$?
    echo "[chain] rc=${rc}"; tail -n 4 "gpurun_out/${log}"
    if [ ${rc} -eq 124 ] || [ ${rc} -eq 137 ]; then echo "[chain] step killed at its limit: stopping"; exit ${rc}; fi
done
exit 0
