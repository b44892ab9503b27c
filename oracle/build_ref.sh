#!/bin/bash
# build_ref.sh — TEST INFRASTRUCTURE, build container only (needs /root/reference).
#
# Compiles the upstream CPU renderer's own translation units with g++ and links
# them with oracle/ref_probe.cpp into oracle/_ref/ref_probe.  Nothing upstream
# enters the repository: the sources are compiled from a scratch copy that is
# deleted afterwards, and oracle/_ref/ is git-ignored.
#
# Why a scratch copy: src/TkMath.h:250/256/262/268 put `static` on out-of-class
# explicit specialisations, which MSVC accepts and g++/clang reject as a hard
# error.  One sed drops those four keywords; nothing else is edited.  Missing
# standard includes are supplied with -include flags, OpenGL declarations come
# from the GLEW header upstream vendors itself (external/include/GL/glew.h;
# -DGLEW_NO_GLU because GL/glu.h is not installed), and the GL entry points,
# which only the viewer's draw functions call, are left unresolved at link time.
# Not compiled: TkViewer.cpp, main.cpp (window loop), TkRenderer.cpp (needs
# GL/glu.h; it also holds CameraInfo::rotate/perspective, whose matrices the
# prober therefore takes as input), and the files upstream's own CMake list
# leaves out (TkSPPM.cpp, core/Renderer.cpp, scattering.cpp).
set -euo pipefail
REF=${REF:-/root/reference}
HERE="$(cd "$(dirname "$0")" && pwd)"
OUT="$HERE/_ref"
[ -d "$REF/src" ] || { echo "build_ref.sh: $REF/src not found (build container only)"; exit 1; }
mkdir -p "$OUT"
S="$(mktemp -d "$OUT/scratch.XXXXXX")"
trap 'rm -rf "$S"' EXIT
cp -r "$REF/src/." "$S/"
chmod -R u+w "$S"
sed -i 's/^\tstatic inline \(f32\|f64\) Math::/\tinline \1 Math::/' "$S/TkMath.h"
# Second edit, for the second prober only (ref_probe2.cpp): get_random_float (src/TkMath.h:312-319) constructs an
# unseeded generator per call, so nothing that calls it can be compared number by number.  It becomes a
# dispatcher: with tk::probe_draw_hook set it returns that function's next number, otherwise it runs
# upstream's own body (renamed, untouched).  ref_probe.cpp never sets the hook: its fixtures do not change.
sed -i 's/^\tinline float get_random_float()$/\textern float (*probe_draw_hook)();\n\tinline float get_random_float_unfed();\n\tinline float get_random_float() { return probe_draw_hook ? probe_draw_hook() : get_random_float_unfed(); }\n\tinline float get_random_float_unfed()/' "$S/TkMath.h"
grep -q 'probe_draw_hook' "$S/TkMath.h" || { echo "build_ref.sh: get_random_float not found where expected"; exit 1; }
INC="-include cstring -include cstdio -include memory -include array -include functional -include cstdint \
 -include algorithm -include math.h -include mutex -include condition_variable -include thread"
FLAGS="-std=c++17 -O3 -w -fPIC -DGLEW_NO_GLU $INC -I$S -I$S/core -I$S/tracer -I$REF/external/include"
mkdir -p "$S/obj"
SRCS=$(cd "$S" && ls *.cpp core/*.cpp tracer/*.cpp | grep -v -e TkViewer -e '^main.cpp' -e TkRenderer -e TkSPPM -e core/Renderer -e scattering)
pids=()
for f in $SRCS; do
    o="$S/obj/$(echo "$f" | tr '/' '_').o"
    g++ $FLAGS -c "$S/$f" -o "$o" &
    pids+=($!)
    if [ ${#pids[@]} -ge 8 ]; then wait "${pids[0]}"; pids=("${pids[@]:1}"); fi
done
for p in "${pids[@]}"; do wait "$p"; done
mkdir -p "$S/probe"
g++ $FLAGS -c "$HERE/ref_probe.cpp" -o "$S/probe/ref_probe.o"
g++ -o "$OUT/ref_probe" "$S"/obj/*.o "$S/probe/ref_probe.o" -lpthread -Wl,--unresolved-symbols=ignore-all
echo "built $OUT/ref_probe"
# the second prober takes upstream's TkSPPM.cpp in by #include (it is not in upstream's build list, see above)
g++ $FLAGS -c "$HERE/ref_probe2.cpp" -o "$S/probe/ref_probe2.o"
g++ -o "$OUT/ref_probe2" "$S"/obj/*.o "$S/probe/ref_probe2.o" -lpthread -Wl,--unresolved-symbols=ignore-all
echo "built $OUT/ref_probe2"
