#!/bin/bash
# Evidence for the gate tests of tests/test_headless_tick.py: the same headless_tick, built against a copy of the drop-in shim in which
# the reference's gate (mesh.cpp:426 / :482, `componentCount == 0 || !isDrawReady(shadowPass)`) is taken out again — what the shim
# did up to commit f2bf690: every system culled and written every frame. The gate cases must FAIL there (exit code 1, "isVisible ...
# was written"), and pass on the shim as it is. Run on a GPU box from the repository root; prints one block per case.
set -u
root=$(pwd)
tmp=$(mktemp -d)
mkdir -p "$tmp/tests/cpp" "$tmp/garden_amd/csrc" "$tmp/garden_amd/lib" "$tmp/oracle" "$tmp/include"
cp -r "$root/tests/cpp/headless_tick.cpp" "$tmp/tests/cpp/"
cp -r "$root/garden_amd/csrc/host" "$tmp/garden_amd/csrc/"
cp "$root"/oracle/*.hpp "$root"/oracle/*.h "$root"/oracle/*.c "$tmp/oracle/"
cp "$root/include/garden_vis.h" "$tmp/include/"
sed -i -e 's/componentCount != 0 \&\& meshSystem->isDrawReady(-1)/true/g' -e 's/componentCount != 0 \&\& meshSystem->isDrawReady(shadowPasses\[s\].index(s))/true/g' \
    "$tmp/garden_amd/csrc/host/gpu_visibility_system.hpp"
grep -c "isDrawReady" "$tmp/garden_amd/csrc/host/gpu_visibility_system.hpp" | sed 's/^/isDrawReady calls left in the patched shim (comments only): /'
cd "$tmp/tests/cpp"
gcc -O2 -march=haswell -ffp-contract=off -fno-fast-math -std=c11 -pthread -c ../../oracle/gv_oracle.c -o gv_oracle.o
gcc -O2 -march=haswell -ffp-contract=off -fno-fast-math -std=c11 -pthread -c ../../oracle/gv_oracle_avx2.c -o gv_oracle_avx2.o
g++ -O2 -std=c++17 -Wno-invalid-offsetof -fno-strict-aliasing -march=haswell -ffp-contract=off -pthread -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include headless_tick.cpp gv_oracle.o gv_oracle_avx2.o \
    -o headless_tick_without_the_gate -L"$root/garden_amd/lib" -lgarden_vis -Wl,-rpath,"$root/garden_amd/lib" -Wl,-rpath,/opt/rocm/lib -L/opt/rocm/lib -lamdhip64 -lm -lpthread || exit 2
for gate in never shadow reverse empty; do
    echo "== --gate $gate, shim WITHOUT the gate (f2bf690's behaviour) =="
    ./headless_tick_without_the_gate --mode both --entities 30000 --mixed --gate $gate --ticks 2; echo "exit code $?"
    echo "== --gate $gate, the shim as it is =="
    "$root/tests/cpp/build/headless_tick" --mode both --entities 30000 --mixed --gate $gate --ticks 2; echo "exit code $?"
done
# ... and the pass NUMBER the gate is asked with (renderShadows, mesh.cpp:809-815: a pass whose prepareShadowRender says no is left out,
# the others keep their numbers): a shim that asks isDrawReady with the pass's POSITION in the list (what it did up to commit 24755f2)
# culls a system in a pass it is not ready for as soon as an earlier pass is left out.
cp "$root/garden_amd/csrc/host/gpu_visibility_system.hpp" "$tmp/garden_amd/csrc/host/gpu_visibility_system.hpp"
sed -i -e 's/isDrawReady(shadowPasses\[s\].index(s))/isDrawReady((int8_t)s)/g' "$tmp/garden_amd/csrc/host/gpu_visibility_system.hpp"
g++ -O2 -std=c++17 -Wno-invalid-offsetof -fno-strict-aliasing -march=haswell -ffp-contract=off -pthread -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include headless_tick.cpp gv_oracle.o gv_oracle_avx2.o \
    -o headless_tick_pass_positions -L"$root/garden_amd/lib" -lgarden_vis -Wl,-rpath,"$root/garden_amd/lib" -Wl,-rpath,/opt/rocm/lib -L/opt/rocm/lib -lamdhip64 -lm -lpthread || exit 2
echo "== --gate shadow --skip-pass 0, shim asking isDrawReady with the pass's POSITION (24755f2's behaviour) =="
./headless_tick_pass_positions --mode both --entities 30000 --mixed --gate shadow --skip-pass 0 --ticks 2; echo "exit code $?"
echo "== --gate shadow --skip-pass 0, the shim as it is =="
"$root/tests/cpp/build/headless_tick" --mode both --entities 30000 --mixed --gate shadow --skip-pass 0 --ticks 2; echo "exit code $?"
rm -rf "$tmp"
