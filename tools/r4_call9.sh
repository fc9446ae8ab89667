O=gpurun_out/r4i
mkdir -p $O
timeout 900 python -m pytest tests/test_msa_gpu.py "tests/test_fullsize_gpu.py::test_config5_msa_full_size_split_weight_mode" "tests/test_fullsize_gpu.py::test_config5_msa_full_size_against_reference_fixture" tests/test_model_gpu.py -q -s > $O/pytest_msa.log 2>&1; echo "rc=$?"; grep -E "passed|failed|FAILED|Error|MSA 4 x|config 5|3B-dims" $O/pytest_msa.log | cut -c1-600 | tail -14
timeout 300 python bench.py --workload msa1b --no-secondary > $O/bench_msa.log 2>&1; grep '^{' $O/bench_msa.log | python -c "import json,sys; r=json.loads(sys.stdin.read()); print('msa1b f16', r['value'], r['ms_per_step'], r.get('parity'))"
timeout 300 python bench.py --workload msa1b --operand f16x2 --no-secondary > $O/bench_msa_f16x2.log 2>&1; grep '^{' $O/bench_msa_f16x2.log | python -c "import json,sys; r=json.loads(sys.stdin.read()); print('msa1b f16x2', r['value'], r['ms_per_step'], r.get('parity'))"; tail -3 $O/bench_msa_f16x2.log | cut -c1-300
