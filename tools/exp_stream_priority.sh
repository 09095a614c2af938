for combo in "0 0" "0 1" "-1 1" "-1 0" "1 -1" "0 -1"; do
  set -- $combo
  echo "main=$1 side=$2: $(ZKHIP_G16_MAIN_PRIORITY=$1 ZKHIP_G16_SIDE_PRIORITY=$2 python tools/bench_groth16.py --steps 8 2>/dev/null | python -c 'import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d["ms_per_proof"], d["verified"])')"
done
