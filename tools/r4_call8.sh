O=gpurun_out/r4h
mkdir -p $O
timeout 1200 python -m pytest tests -m gpu -q -s > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -5 $O/pytest_gpu.log
grep -E "FAILED|Error" $O/pytest_gpu.log | head -20
grep -E "rel|err|L2|argmax|logit|floor|MSA \(|dims|worst" $O/pytest_gpu.log | grep -v "^tests/" > $O/gpu_tests_parity_lines.txt
