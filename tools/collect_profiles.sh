#!/bin/bash
# Evidence run of a round (on the GPU box, from the repository root; R=r06 by default): bench line + sidecar, rocprofv3 kernel stats, PMC
# passes, sweeps.  Everything lands in gpurun_out/ (copied into profiles/ afterwards).  QUICK=1 skips the sweeps and micro-benchmarks.
set -u
root=${GRAFT_REPO_ROOT:-/root/repo}
cd "$root"
mkdir -p gpurun_out
export TMPDIR=/tmp
R=${R:-r06}
python3 bench.py > gpurun_out/${R}_bench_final.json 2> gpurun_out/${R}_bench_final.err
cp bench_detail.json gpurun_out/${R}_bench_final_detail.json
# the driver's own command line (20 timed steps after 5 warm-up steps)
python3 bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/${R}_bench_driver_flags.json 2> /dev/null
cp bench_detail.json gpurun_out/${R}_bench_driver_flags_detail.json
# kernel trace + stats of the MSM / NTT legs of the same command
rm -rf /tmp/rp && ( cd /tmp && rocprofv3 --kernel-trace --stats -d /tmp/rp -o msm --output-format csv -- python3 "$root/bench.py" --no-cpu-baseline --no-groth16 --no-kzg --no-pmc --no-verify --no-two-in-flight > /tmp/rp.log 2>&1 )
cp $(find /tmp/rp -name '*kernel_stats.csv' | head -1) gpurun_out/${R}_rocprofv3_kernel_stats_msm_bench.csv 2>/dev/null
# the same for a whole Groth16 proof (both streams) and the 50-column KZG commit
rm -rf /tmp/rp2 && ( cd /tmp && rocprofv3 --kernel-trace --stats -d /tmp/rp2 -o g16 --output-format csv -- python3 "$root/tools/bench_groth16.py" --steps 4 > /tmp/rp2.log 2>&1 )
cp $(find /tmp/rp2 -name '*kernel_stats.csv' | head -1) gpurun_out/${R}_rocprofv3_kernel_stats_groth16.csv 2>/dev/null
rm -rf /tmp/rp3 && ( cd /tmp && rocprofv3 --kernel-trace --stats -d /tmp/rp3 -o kzg --output-format csv -- python3 "$root/tools/bench_kzg.py" > /tmp/rp3.log 2>&1 )
cp $(find /tmp/rp3 -name '*kernel_stats.csv' | head -1) gpurun_out/${R}_rocprofv3_kernel_stats_kzg.csv 2>/dev/null
# the placeholder legs (round, lookup, permutation): the lines and their kernel table
python3 tools/run_placeholder_legs.py > gpurun_out/${R}_placeholder_legs.json 2>/dev/null
rm -rf /tmp/rp4 && ( cd /tmp && rocprofv3 --kernel-trace --stats -d /tmp/rp4 -o ph --output-format csv -- python3 "$root/tools/run_placeholder_legs.py" > /tmp/rp4.log 2>&1 )
cp $(find /tmp/rp4 -name '*kernel_stats.csv' | head -1) gpurun_out/${R}_rocprofv3_kernel_stats_placeholder_legs.csv 2>/dev/null
# PMC: one counter group per pass (FETCH_SIZE / WRITE_SIZE in passes of their own), over the MSM + NTT workload of tools/pmc_child.py
tools/pmc_collect.sh gpurun_out/${R}_pmc_msm_ntt.json tools/pmc_child.py 20 > /dev/null 2>&1
PMC_GROUPS="GRBM_GUI_ACTIVE;SQ_BUSY_CYCLES SQ_WAVES" tools/pmc_collect.sh gpurun_out/${R}_pmc_clock.json tools/pmc_child.py 20 > /dev/null 2>&1
tools/pmc_collect.sh gpurun_out/${R}_pmc_g2_msm.json tools/msm_g2_once.py > /dev/null 2>&1
# the grand-product kernels on their own
rm -rf /tmp/rp5 && ( cd /tmp && rocprofv3 --kernel-trace --stats -d /tmp/rp5 -o gp --output-format csv -- python3 "$root/tools/perm_profile.py" > /tmp/rp5.log 2>&1 )
cp $(find /tmp/rp5 -name '*kernel_stats.csv' | head -1) gpurun_out/${R}_rocprofv3_kernel_stats_grand_products.csv 2>/dev/null
# round 6: the realistic gate argument (fused against per-term) and its kernel table
python3 tools/bench_gate_argument.py > gpurun_out/${R}_gate_argument.json 2>/dev/null
rm -rf /tmp/rp6 && ( cd /tmp && rocprofv3 --kernel-trace --stats -d /tmp/rp6 -o gate --output-format csv -- python3 "$root/tools/bench_gate_argument.py" > /tmp/rp6.log 2>&1 )
cp $(find /tmp/rp6 -name '*kernel_stats.csv' | head -1) gpurun_out/${R}_rocprofv3_kernel_stats_gate_argument.csv 2>/dev/null
python3 tools/call_overhead.py > gpurun_out/${R}_call_overhead.txt 2>/dev/null
for seed in 1 2; do timeout 400 python3 tests/fuzz_gpu.py --seconds 300 --seed $seed 2>&1 | tail -3; done > gpurun_out/${R}_fuzz.txt
if [ -n "${QUICK:-}" ]; then ls -la gpurun_out | tail -20; exit 0; fi
python3 tools/msm_sweep.py 2>/dev/null | tail -1 > gpurun_out/${R}_msm_size_sweep.json
python3 tools/shard_emulation.py 20 2>/dev/null | tail -1 > gpurun_out/${R}_shard_emulation.json
python3 tools/shard_emulation.py --proof 20 2>/dev/null | tail -1 > gpurun_out/${R}_shard_emulation_proof.json
python3 tools/bench_groth16.py --steps 6 2>/dev/null | tail -1 > gpurun_out/${R}_groth16_2p20_shim.json
python3 tools/bench_groth16.py --steps 6 --domain basic 2>/dev/null | tail -1 > gpurun_out/${R}_groth16_2p20_basic_domain_shim.json
./tools/kzg_shim_bench 20 50 10 > gpurun_out/${R}_kzg_shim_bench.txt 2>&1
./tools/lpc_shim_bench 20 16 16 > gpurun_out/${R}_lpc_shim_bench.txt 2>&1
python3 tools/bench_kzg.py 2>/dev/null | tail -1 > gpurun_out/${R}_kzg_commit_50x2p20.json
python3 tools/bench_ntt.py 2>/dev/null | tail -1 > gpurun_out/${R}_ntt_2p22x8.json
( python3 tools/bench_ecntt.py 12 16 18 20; python3 tools/bench_ecntt.py 12 14 --g2; python3 tools/bench_ecntt.py 16 --curve1 ) > gpurun_out/${R}_ecntt.txt 2>/dev/null
python3 tools/groth16_two_provers.py 2>/dev/null | tail -1 > gpurun_out/${R}_groth16_two_provers.json
( cd tools && ./microbench2 ) > gpurun_out/${R}_microbench2_valu_wallclock.txt 2>&1
( cd tools && ./mulbench4 | grep -v "^CHECK" ) > gpurun_out/${R}_mulbench4_asm_vs_cpp_13x30.txt 2>&1
ZKHIP_GEN_PHASES=1 python3 tools/bench_groth16.py --steps 2 2>&1 | grep "generator phase" > gpurun_out/${R}_generator_phases.txt
python3 tools/bench_groth16.py --steps 6 --curve 1 2>/dev/null | tail -1 > gpurun_out/${R}_groth16_2p20_bn254_shim.json
ls -la gpurun_out | tail -20
