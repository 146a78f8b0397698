# build_variant.sh NAME -D... : librsis_hip.so with conv_wgrad_bf16.hip compiled under extra defines -> rsis_amd/lib/exp/librsis_NAME.so
set -e
name=$1; shift
cd $(dirname $0)/../../rsis_amd/csrc
mkdir -p build/exp ../lib/exp
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-gpu-rdc -Wall -Wno-unused-function "$@" -c conv_wgrad_bf16.hip -o build/exp/$name.o
objs=$(ls build/*.o | grep -v conv_wgrad_bf16.o)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../lib/exp/librsis_$name.so $objs build/exp/$name.o -ldl
echo built $name
