# Diagnostic build + run of tools/gemm_stamps.py (in-kernel stamps of the persistent GEMM's tile anatomy).  Leaves the diagnostic library under
# gpurun_out/libavllm_stamps.so and rebuilds nothing in place: the shipped library is untouched.
set -e
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; cd $R
mkdir -p gpurun_out/stamp_obj
for f in audio-visual-llm_amd/csrc/*.hip; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -Wno-unused-result -I include -DAVLLM_GEMM_STAMPS -c $f -o gpurun_out/stamp_obj/$(basename $f .hip).o &
done; wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o gpurun_out/libavllm_stamps.so gpurun_out/stamp_obj/*.o
AVLLM_LIB_PATH=$R/gpurun_out/libavllm_stamps.so python3 tools/gemm_stamps.py
