export TMPDIR=/tmp
for lds in 6144 8192 9216 10240 11264 12288 14336; do
  for i in 1 2; do echo "lds=$lds $(MJHIP_LDS_BYTES=$lds python bench.py --no-cpu-baseline 2>&1 | tail -1 | cut -c66-86)"; done
done
