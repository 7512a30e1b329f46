# Cache-policy bits on the operand DMA (global_load_lds_dwordx4) of the persistent GEMM: one library per policy, built HERE (no GPU needed) into
# tmp_pol/, then `tools/gemm_dma_policy.sh run` on the GPU box times tools/gemm_bench.py with each.  tmp_pol/ is scratch: delete it afterwards.
set -e
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; cd $R
if [ "$1" = "run" ]; then
  for f in tmp_pol/libavllm_*.so; do
    echo "== $f"; AVLLM_LIB_PATH=$R/$f python3 tools/gemm_bench.py 2>&1 | grep -v amdgpu.ids
  done
  exit 0
fi
mkdir -p tmp_pol
i=0
for pol in "" " sc0" " sc1" " sc0 sc1" " nt" " sc1 nt"; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -Wno-unused-result -I include "-DAVLLM_DMA_POLICY=\"$pol\"" -c audio-visual-llm_amd/csrc/gemm.hip -o tmp_pol/gemm_$i.o
  objs=$(ls audio-visual-llm_amd/csrc/build/*.o | grep -v "/gemm.o")
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o "tmp_pol/libavllm_$i$(echo "$pol" | tr ' ' '_').so" tmp_pol/gemm_$i.o $objs
  i=$((i+1))
done
rm -f tmp_pol/*.o; ls -la tmp_pol
