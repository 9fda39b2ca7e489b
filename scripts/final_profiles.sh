# Every profile of profiles/rNN_* that quotes the final code, in one gpurun call (one box): bash scripts/final_profiles.sh > gpurun_out/fin_all.log 2>&1
# then copy gpurun_out/fin_* / prof_fin* / traffic / frangi_fin / gaps_fin into profiles/ (profiles/README.md says which file is which)
set -x
cd $GRAFT_REPO_ROOT
for i in 1 2 3; do python bench.py > gpurun_out/fin_bench_$i.json 2> gpurun_out/fin_bench_$i.err || exit 1; done
PNR_BENCH_OPTS=groups=1 bash scripts/prof_traffic.sh > gpurun_out/fin_traffic.log 2>&1 || exit 1
bash scripts/profile_bench.sh fin2 --steps 2 --no-cpu-baseline --no-extra > gpurun_out/fin_prof2.log 2>&1 || exit 1
PNR_BENCH_OPTS=groups=1 bash scripts/profile_bench.sh fin1 --steps 2 --no-cpu-baseline --no-extra > gpurun_out/fin_prof1.log 2>&1 || exit 1
bash scripts/prof_frangi.sh fin 1024 > gpurun_out/fin_frangi.log 2>&1 || exit 1
python scripts/cli_wall.py 1024 > gpurun_out/fin_cli_wall.json 2> gpurun_out/fin_cli_wall.err || exit 1
python scripts/emulate_ranks.py --worlds 1,2,4,8 > gpurun_out/fin_emulate.txt 2>&1 || exit 1
bash scripts/prof_gaps.sh fin "" > gpurun_out/fin_gaps.log 2>&1 || exit 1
[ -x scripts/probes/hip_init ] || /opt/rocm/bin/hipcc --offload-arch=gfx950 -O2 -o scripts/probes/hip_init scripts/probes/hip_init.cpp -ldl
for i in 1 2 3; do ./scripts/probes/hip_init >> gpurun_out/fin_hip_init.txt; ./scripts/probes/hip_init pnr_amd/libpnr_hip.so >> gpurun_out/fin_hip_init.txt; done
echo ALLDONE
