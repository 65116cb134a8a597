set -e
EGOTAP_LIB=$GRAFT_REPO_ROOT/egotap_amd/libegotap_flat.so timeout -k 10 500 python bench.py --no-cpu-baseline > gpurun_out/ab_flat.json 2>gpurun_out/ab_flat.err
timeout -k 10 500 python bench.py --no-cpu-baseline > gpurun_out/ab_sbase.json 2>gpurun_out/ab_sbase.err
for i in 1 2; do
EGOTAP_LIB=$GRAFT_REPO_ROOT/egotap_amd/libegotap_flat.so timeout -k 10 200 python tools/hm_bf16_probe.py 256 64 bf16 | tail -1
timeout -k 10 200 python tools/hm_bf16_probe.py 256 64 bf16 | tail -1
done
