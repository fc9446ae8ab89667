#!/bin/bash
# whole -m gpu suite in both engine modes (fold default, ESM_AMD_LN_FOLD=0), contract lines extracted
set -u
O=gpurun_out/${1:-r6s}
mkdir -p $O
T0=$(date +%s)
timeout 900 python -m pytest tests -m gpu -q -s -p no:cacheprovider > $O/pytest_gpu.log 2>&1; echo "pytest fold rc=$? $(( $(date +%s) - T0 )) s"; tail -3 $O/pytest_gpu.log
grep -E "contract |FAILED|Error" $O/pytest_gpu.log > $O/contract_lines.txt
ESM_AMD_LN_FOLD=0 timeout 900 python -m pytest tests -m gpu -q -s -p no:cacheprovider > $O/pytest_gpu_plain.log 2>&1; echo "pytest plain rc=$? $(( $(date +%s) - T0 )) s"; tail -3 $O/pytest_gpu_plain.log
grep -E "contract |FAILED|Error" $O/pytest_gpu_plain.log > $O/contract_lines_plain.txt
