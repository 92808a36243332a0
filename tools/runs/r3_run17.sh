set -e
mkdir -p gpurun_out
cd madqp_jl_amd/csrc
hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -ffp-contract=off -DMADQP_BATCH_STAMPS -c batch.hip -o /tmp/batch_st.o
hipcc --offload-arch=gfx950 -shared -fPIC -o /tmp/libst.so ctx.o gemm_f64.o chol.o gemv.o vec_kernels.o gen.o kkt.o mpc.o /tmp/batch_st.o sparse.o coo.o dist.o -ldl
cd ../..
cp madqp_jl_amd/libmadqp_hip.so /tmp/lib_orig.so
cp /tmp/libst.so madqp_jl_amd/libmadqp_hip.so
python tools/batch_stamps.py 1024 > gpurun_out/r17_stamps.txt 2>&1; cp /tmp/lib_orig.so madqp_jl_amd/libmadqp_hip.so
cat gpurun_out/r17_stamps.txt
