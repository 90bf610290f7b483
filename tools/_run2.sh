set -u
mkdir -p gpurun_out
for i in 1 2; do
tools/variant_probe.sh 20000 64 16 1 2>&1 | grep -v "^\[W\|amdgpu.ids"
done
