cd $GRAFT_REPO_ROOT
for e in 0 1 2 4 8 15; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Iinclude -Ifactorized_amd/csrc -fno-slp-vectorize -DMFM_EXP=$e -c factorized_amd/csrc/lstm_seq_small.hip -o factorized_amd/csrc/build/lstm_seq_small.o 2>/dev/null
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC factorized_amd/csrc/build/*.o -o factorized_amd/libmfm_hip.so
  echo "=== EXP $e"
  BENCH_SEQ_PATHS=small BENCH_SEQ_BWD=1 timeout 120 python scripts/bench_seq.py 120 32 2>&1 | grep -v amdgpu.ids
done
