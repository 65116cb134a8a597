for i in 1 2 3; do
  echo "flat:"; EGOTAP_LIB=$GRAFT_REPO_ROOT/egotap_amd/libegotap_flat.so timeout -k 10 300 python bench.py --steps 10 --warmup 3 --lift-only --no-fast-mode --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['frac'])"
  echo "sbase:"; timeout -k 10 300 python bench.py --steps 10 --warmup 3 --lift-only --no-fast-mode --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['frac'])"
done
