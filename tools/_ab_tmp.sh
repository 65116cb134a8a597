for i in 1 2 3; do
EGOTAP_LIB=$GRAFT_REPO_ROOT/egotap_amd/libegotap_flat.so timeout -k 10 200 python tools/hm_bf16_probe.py 256 64 bf16 | tail -1
timeout -k 10 200 python tools/hm_bf16_probe.py 256 64 bf16 | tail -1
done
