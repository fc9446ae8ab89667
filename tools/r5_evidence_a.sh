#!/bin/bash
# Round-5 evidence pass A (one box) for the library in the tree: rocprofv3 kernel stats + PMC passes (fold default at B = 64 and
# B = 4, the plain mode at B = 64, 3B contacts, MSA-1b), the first-K-tile fix stamped per tile (tools/bench_gemm9.py on the library
# and on its -DESMK_G9_TIE1=0 build), the whole -m gpu suite.  Output: gpurun_out/r5a/
set -u
O=gpurun_out/r5a
mkdir -p $O
T0=$(date +%s)
LIB=esm_amd/lib/libesmk.so
timeout 600 python -m pytest tests -m gpu -q -s -p no:cacheprovider > $O/pytest_gpu.log 2>&1; echo "pytest rc=$? $(( $(date +%s) - T0 )) s"; tail -3 $O/pytest_gpu.log
grep -E "^contract |FAILED|Error|degenerate" $O/pytest_gpu.log > $O/contract_lines.txt
bash tools/profile_bench.sh r5a/prof_650m esm2_650m > $O/profile_650m.log 2>&1; echo "prof 650m $(( $(date +%s) - T0 )) s"
bash tools/profile_bench.sh r5a/prof_650m_b4 esm2_650m 4 1 > $O/profile_650m_b4.log 2>&1
bash tools/profile_bench.sh r5a/prof_650m_plain esm2_650m 64 0 > $O/profile_650m_plain.log 2>&1; echo "prof plain $(( $(date +%s) - T0 )) s"
bash tools/profile_bench.sh r5a/prof_3b esm2_3b_contacts > $O/profile_3b.log 2>&1
bash tools/profile_bench.sh r5a/prof_msa msa1b > $O/profile_msa.log 2>&1; echo "prof msa $(( $(date +%s) - T0 )) s"
# first K tile: cycles per K tile / epilogue / seam of the four layer shapes, the library and its no-tie build, same box
timeout 300 python tools/bench_gemm9.py --no-vendor --no-check --rounds 3 > $O/gemm9_tie.log 2>&1
cp $LIB /tmp/lib_keep.so; cp esm_amd/lib/variants/libesmk_notie.so $LIB
timeout 300 python tools/bench_gemm9.py --no-vendor --no-check --rounds 3 > $O/gemm9_notie.log 2>&1
cp /tmp/lib_keep.so $LIB
echo "total $(( $(date +%s) - T0 )) s"
ls $O/prof_*/pmc_summary.json
