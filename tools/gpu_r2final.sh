# end-of-round measurement of the committed build (runs on the GPU box):
# GPU tests, smoke, the default and the driver-config profile (rocprofv3 stats + PMC passes), tail statistics,
# PCIe-inclusive rate
set -x
export TMPDIR=/tmp
mkdir -p gpurun_out/r2final
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/r2final/pytest_gpu.log 2>&1; tail -2 gpurun_out/r2final/pytest_gpu.log
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
bash tools/gpu_profile.sh r02d > gpurun_out/r2final/profile_default.log 2>&1; tail -3 gpurun_out/prof_r02d/pmc_summary.txt
bash tools/gpu_profile.sh r02d_steps20 --steps 20 --warmup 5 > gpurun_out/r2final/profile_steps20.log 2>&1; tail -3 gpurun_out/prof_r02d_steps20/pmc_summary.txt
python tools/tail_stats.py 100 6 > gpurun_out/r2final/tail_stats.txt 2>&1; tail -12 gpurun_out/r2final/tail_stats.txt
python tools/pcie_rate.py 4096 100 5 > gpurun_out/r2final/pcie_rate.txt 2>&1; cat gpurun_out/r2final/pcie_rate.txt
