O=gpurun_out/r4g
mkdir -p $O
df -h /dev/shm | tail -1; nproc; free -g | head -2
for spec in "--crc32 --gpu-ms 0" "--gpu-ms 0" "--gpu-ms 0 --writer-threads 16" "--gpu-ms 0 --writer-threads 4" "--gpu-ms 100" "--gpu-ms 0 --out-root /tmp"; do
  timeout 300 python tools/bench_extract_hosts.py --world 8 --seqs-per-rank 512 $spec 2>&1 | grep '^{' | tee -a $O/extract_hosts.log
done
timeout 200 python tools/bench_extract_hosts.py --world 1 --seqs-per-rank 512 --gpu-ms 0 2>&1 | grep '^{' | tee -a $O/extract_hosts.log
