# build: cd tools/lt_probe && hipcc --offload-arch=gfx950 -O2 -o lt_probe lt_probe.cpp -lhipblaslt
# shapes (column-major as torch.bmm passes them): fwd4, dcols4, dW4 (S=3), fwd3, dcols3, dW3, fwd2, dcols2, dW2 (S=8)
P=tools/lt_probe/lt_probe
R=6144
run() { $P "$@" | tail -n 1; }
for lib in /opt/rocm/lib $(python -c "import torch,os;print(os.path.join(os.path.dirname(torch.__file__),'lib'))"); do
echo "== hipBLASLt from $lib"; export LD_LIBRARY_PATH=$lib
run 1024 $R 5120 N N 1024 5120 1024 5 $((5120*1024)) $((R*5120)) $((R*1024))
run 5120 $R 3072 T N 3072 3072 5120 5 $((5120*3072)) $((R*3072)) $((R*5120))
run 5120 1024 2048 N T 5120 3072 5120 15 $((2048*5120)) $((2048*3072)) $((1024*5120))
run 1024 $R 2560 N N 1024 2560 1024 5 $((2560*1024)) $((R*2560)) $((R*1024))
run 2560 $R 3072 T N 3072 3072 2560 5 $((2560*3072)) $((R*3072)) $((R*2560))
run 2560 1024 $R N T 2560 3072 2560 5 $((R*2560)) $((R*3072)) $((1024*2560))
run 512 18432 640 N N 512 640 512 5 $((640*512)) $((18432*640)) $((18432*512))
run 640 18432 1536 T N 1536 1536 640 5 $((640*1536)) $((18432*1536)) $((18432*640))
run 640 512 2304 N T 640 1536 640 40 $((2304*640)) $((2304*1536)) $((512*640))
done
