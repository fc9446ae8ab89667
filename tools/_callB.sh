O=gpurun_out/r3B
mkdir -p $O
LIB=esm_amd/lib/libesmk.so
cp $LIB /tmp/libesmk_new.so
run() { tag=$1; shift; timeout 300 python bench.py "$@" --no-cpu-baseline > $O/ab_$tag.log 2>&1; python -c "import json; r=json.loads([l for l in open('$O/ab_$tag.log') if l.startswith('{')][-1]); print('$tag', r['value'], r['ms_per_step'], r['library']['src_hash'], {k: v['ms_per_step'] for k, v in r['kernel_classes'].items() if v['ms_per_step'] > 0.7})"; }
run new_650m
cp esm_amd/lib/libesmk_prev.so $LIB
run old_650m
run old_b4 --batch 4
run old_b16 --batch 16
run old_3b --workload esm2_3b_contacts --steps 4
run old_msa --workload msa1b
cp /tmp/libesmk_new.so $LIB
run new2_650m
run new_b4 --batch 4
run new_b16 --batch 16
run new_3b --workload esm2_3b_contacts --steps 4
run new_msa --workload msa1b
ESMK_GEMM_IMPL=8 run new_impl8_650m
