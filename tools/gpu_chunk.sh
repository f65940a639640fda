export TMPDIR=/tmp
for i in 1 2 3; do
  for c in 100 250 500; do echo "chunk=$c $(python bench.py --no-cpu-baseline --chunk $c 2>&1 | tail -1 | cut -c66-86)"; done
done
for i in 1 2; do
  for c in 100 1000; do echo "steps=1000 chunk=$c $(python bench.py --no-cpu-baseline --steps 1000 --chunk $c 2>&1 | tail -1 | cut -c66-86)"; done
done
