mkdir -p gpurun_out/ab
for r in 1 2 3; do
for v in base hoist; do
VINGS_HIP_LIB=$PWD/tools/experiments/ab/lib_$v.so timeout -k 10 120 python bench.py --no-cpu-baseline --no-side-runs --steps 20 > gpurun_out/ab/${v}_$r.json 2> gpurun_out/ab/${v}_$r.err || exit 1
done; done
echo ab-ok
