#!/bin/bash
# A/B of two builds of libesmk.so inside ONE gpurun call (boxes differ by +-3 %, so variants are only ever compared
# inside a call).  Before the call: build the candidate into esm_amd/lib/libesmk.so and keep the baseline as
# esm_amd/lib/libesmk_prev.so (both travel with the snapshot; *.so is git-ignored).  Each bench line carries the
# source hash of the library it ran on (`library.src_hash`).
# usage: tools/ab_two_libraries.sh <outdir-under-gpurun_out> [workloads...]      default workloads: 650m 3b msa b4
set -u
OUT=gpurun_out/${1:-ab}
shift
WL=${@:-650m 3b msa b4}
mkdir -p $OUT
LIB=esm_amd/lib/libesmk.so
[ -f esm_amd/lib/libesmk_prev.so ] || { echo "esm_amd/lib/libesmk_prev.so missing"; exit 2; }
cp $LIB /tmp/libesmk_new.so
run() {  # $1 = tag
  for w in $WL; do
    case $w in
      650m) a="" ;;
      3b)   a="--workload esm2_3b_contacts --steps 4" ;;
      msa)  a="--workload msa1b" ;;
      b4)   a="--batch 4" ;;
      b16)  a="--batch 16" ;;
      *)    a="$w" ;;
    esac
    timeout 300 python bench.py $a --no-cpu-baseline > $OUT/$1_$w.log 2>&1
    grep '^{' $OUT/$1_$w.log | python -c "
import sys, json
r = json.loads(sys.stdin.read())
print('$1 $w', r['value'], r['unit'], r['ms_per_step'], 'ms/step', r.get('library', {}).get('src_hash'),
      {k: v['ms_per_step'] for k, v in r.get('kernel_classes', {}).items() if 'gemm' in k or 'attention' in k or 'contact' in k or 'probs' in k})" 2>/dev/null
  done
}
run new                                  # candidate, cold box
cp esm_amd/lib/libesmk_prev.so $LIB; run old
cp /tmp/libesmk_new.so $LIB; run new2    # candidate again: brackets the baseline in time
# environment-switched variants of the candidate (off by default): q/k and v projections on two streams
for w in b4 b16 msa; do
  case $w in b4) a="--batch 4" ;; b16) a="--batch 16" ;; msa) a="--workload msa1b" ;; esac
  ESMK_QKV_FORK=1 timeout 300 python bench.py $a --no-cpu-baseline > $OUT/fork_$w.log 2>&1
  timeout 300 python bench.py $a --no-cpu-baseline > $OUT/nofork_$w.log 2>&1
  for t in fork nofork; do grep '^{' $OUT/${t}_$w.log | python -c "
import sys, json
r = json.loads(sys.stdin.read()); print('$t $w', r['value'], r['ms_per_step'])" 2>/dev/null; done
done
