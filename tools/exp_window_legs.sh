for c in 17 18 19 20 21; do
  echo "== c=$c"
  ZKHIP_MSM_WINDOW_BITS=$c python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-ntt --no-pmc 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('msm', d['ms_per_step'], d['verified'], 'g16', d['groth16']['ms_per_proof'], d['groth16']['verified'], 'g16m20', d['groth16_m2p20']['ms_per_proof'], 'kzg', d['kzg']['ms_per_commit'], d['kzg']['opening_proof_ms'], d['kzg']['verified'])"
done
