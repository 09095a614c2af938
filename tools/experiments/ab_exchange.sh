for i in 1 2 3; do
for f in "" "--sync-exchange"; do
python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29777 bench.py --gpus 1 --steps 40 --warmup 5 --force-dist $f --no-cpu-baseline --no-pmc --no-groth16 --no-kzg --no-ntt --no-verify 2>/dev/null | grep "^{" | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$f', d['value'], d['ms_per_step'], d['dist']['exchange'][:12])"
done; done
