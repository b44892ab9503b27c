"""Generates tests/golden/ref_*.npz from the upstream renderer itself — TEST
INFRASTRUCTURE, build container only (needs /root/reference).

    bash oracle/build_ref.sh                    # compiles upstream's sources + oracle/ref_probe.cpp, ref_probe2.cpp
    python oracle/make_ref_golden.py            # runs the probers, packs their arrays (every reproducible fixture)
    python oracle/make_ref_golden.py --only none --renders   # the whole-render image fixtures (minutes, not reproducible)

The fixtures are data (inputs and the outputs upstream's functions gave for
them); the prober's source is oracle/ref_probe.cpp.  Inputs that upstream does
not produce itself in the headless build are made here and recorded in the
fixtures: the camera matrices (CameraInfo::rotate/perspective live in a file
that needs GL/glu.h, so the prober takes them as input) and the photons of the
traceTile fixture (one photon pass of the CPU oracle on cornell_sphere.scene;
any photons on the scene's surfaces would do).
"""
import math
import os
import subprocess
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")
PROBE = os.path.join(ROOT, "oracle", "_ref", "ref_probe")

# fixture -> prefixes of the prober's arrays that go into it
GROUPS = {
    "ref_lowdiscrepancy.npz": ["radinv_", "hash_", "perm_", "strat_", "sampling_"],
    "ref_knn.npz": ["dirtab", "knn100_", "knn50_", "knnties_"],
    "ref_materials.npz": ["mat_"],
    "ref_camera.npz": ["camera", "cam_in_"],
    "ref_rays_box.npz": ["rays_box_"],
    "ref_rays_sphere.npz": ["rays_sphere_"],
    "ref_rays_dragon.npz": ["rays_dragon_"],
    "ref_trace_tile.npz": ["tile_"],
    "ref_nee.npz": ["nee_"],   # inputs: the first rays of the ref_rays_* fixtures of the same scene
}


def camera_to_screen(cam, width, height):
    """CameraInfo::perspective (src/TkRenderer.cpp:358-376) in float32 steps."""
    f = np.float32
    fov_rad = f(cam.fov_degrees) * (f(math.pi) / f(180.0))
    h = f(1.0) / f(math.tan(float(fov_rad * f(0.5))))
    w = h * f(height) / f(width)
    fa, ne = f(cam.far_clip), f(cam.near_clip)
    m = np.zeros(16, np.float32)
    m[0] = w
    m[5] = h
    m[10] = -fa / (fa - ne)
    m[11] = -fa * ne / (fa - ne)
    m[14] = -1.0
    return m


SEED = 0x5EED
PROBE2 = os.path.join(ROOT, "oracle", "_ref", "ref_probe2")
SCENES_DIR = os.path.join(ROOT, "assets", "scenes")

# The one edit build_ref.sh makes for the second prober, recorded in the fixtures that depend on it.
DRAW_PATCH = ("src/TkMath.h:312-319, in the scratch copy only: `inline float get_random_float()` becomes "
              "`extern float (*probe_draw_hook)(); inline float get_random_float_unfed(); inline float get_random_float() "
              "{ return probe_draw_hook ? probe_draw_hook() : get_random_float_unfed(); }` followed by the original body "
              "under the name get_random_float_unfed.  ref_probe2 sets the hook to hand out, call by call, one row of a draw "
              "table per path: oracle_py.photon_draws(0x5EED, kind, H0, n, per) / pt_draws(0x5EED, sample, pixels, per), the "
              "numbers the CPU oracle and the HIP kernels use at the same places.")

# fixtures of the second prober -> prefixes; arrays ending in _draws are inputs that the oracle reproduces and are not kept
GROUPS2 = {
    "ref_photon_paths.npz": ["paths_"],
    "ref_pt_li.npz": ["li_"],
    "ref_hsppm.npz": ["hsppm_"],
}
PATH_SCENES = [("cornell_box.scene", "box", 2048), ("cornell_sphere.scene", "sphere", 2048), ("cornell_dragon.scene", "dragon", 1024)]
LI_SCENES = [("cornell_box.scene", "box"), ("cornell_sphere.scene", "sphere")]
HSPPM = ("cornell_sphere.scene", "sphere", 20000, 0.15)      # photons per iteration, initial radius
RENDERS = [("cornell_box.scene", "box", 0), ("cornell_sphere.scene", "sphere", 5000)]   # caustic map size
RENDER_SHAPE = dict(W=200, H=200, recurse=4, gsize=20000, runs=8, threads=1)


def write_camera(scene, ind):
    c2w, _ = scene.camera_matrices(48, 48)
    c2s = camera_to_screen(scene.camera, 48, 48)   # square images: the same matrix at every size
    np.concatenate([c2w, c2s]).astype("<f4").tofile(os.path.join(ind, "camera.f32"))
    return c2w, c2s


def collect(outd):
    return {fn[:-4]: np.load(os.path.join(outd, fn), allow_pickle=False) for fn in sorted(os.listdir(outd)) if fn.endswith(".npy")}


def first_prober(scene, golden_in):
    """ref_probe: the fixtures of GROUPS.  The photons of the traceTile fixture are inputs: the committed fixture's
    own arrays when there is one (so that a rerun reproduces it), a photon pass of the oracle otherwise."""
    from oracle import oracle_py as O
    with tempfile.TemporaryDirectory() as tmp:
        ind, outd = os.path.join(tmp, "in"), os.path.join(tmp, "out")
        os.makedirs(ind)
        os.makedirs(outd)
        c2w, c2s = write_camera(scene, ind)
        extra = {}
        have = os.path.join(golden_in, "ref_trace_tile.npz")
        if os.path.exists(have):
            old = np.load(have, allow_pickle=False)
            for tag in ("g", "c"):
                for part in ("pos", "power", "theta", "phi"):
                    extra[f"tile_{tag}_{part}"] = old[f"tile_{tag}_{part}"]
        else:
            ng, nc = 20000, 3000
            osc = O.OracleScene(scene, 48, 48)
            orr = O.OracleSppm(osc, scene.sppm_param(globalSize=ng, causticSize=nc, numRecurse=4), scene.tracer.max_depth,
                               scene.tracer.russian_roulette, SEED, threads=4)
            orr.photon_pass(0)
            for which, tag, cap in ((0, "g", ng), (1, "c", nc)):
                pos, power, th, ph, _ = orr.photons(which, cap)
                assert len(pos) == cap
                extra[f"tile_{tag}_pos"], extra[f"tile_{tag}_power"] = pos, power
                extra[f"tile_{tag}_theta"], extra[f"tile_{tag}_phi"] = th, ph
            orr.close()
            osc.close()
        for tag in ("g", "c"):
            extra[f"tile_{tag}_pos"].astype("<f4").tofile(os.path.join(ind, f"{tag}_pos.f32"))
            extra[f"tile_{tag}_power"].astype("<f4").tofile(os.path.join(ind, f"{tag}_pow.f32"))
            np.stack([extra[f"tile_{tag}_theta"], extra[f"tile_{tag}_phi"]], 1).astype(np.uint8).tofile(os.path.join(ind, f"{tag}_dir.u8"))
        extra["cam_in_c2w"], extra["cam_in_c2s"] = c2w, c2s
        cfg = scene.camera
        extra["cam_in_config"] = np.array([cfg.fov_degrees, cfg.near_clip, cfg.far_clip, cfg.theta, cfg.phi, cfg.r], np.float32)
        subprocess.run([PROBE, ind, outd], cwd=SCENES_DIR, check=True)
        arrays = dict(extra)
        arrays.update(collect(outd))
    return arrays


def second_prober(scene, which):
    """ref_probe2 with fed draws: light-path loops, PathTracer::Li, tk::SPPM (`which`: the fixtures wanted)."""
    from oracle import oracle_py as O
    arrays = {}
    with tempfile.TemporaryDirectory() as tmp:
        ind, outd = os.path.join(tmp, "in"), os.path.join(tmp, "out")
        os.makedirs(ind)
        os.makedirs(outd)
        write_camera(scene, ind)
        if "ref_photon_paths.npz" in which:
            for file, tag, n in PATH_SCENES:
                O.photon_draws(SEED, 0, 0, n, 65).tofile(os.path.join(ind, f"draws_{tag}_global.f32"))
                O.photon_draws(SEED, 1, 0, n, 65).tofile(os.path.join(ind, f"draws_{tag}_caustic.f32"))
                subprocess.run([PROBE2, "paths", ind, outd, file, tag], cwd=SCENES_DIR, check=True)
        if "ref_pt_li.npz" in which:
            for file, tag in LI_SCENES:
                O.pt_draws(SEED, 0, 32 * 32, 20).tofile(os.path.join(ind, f"draws_{tag}_li.f32"))
                subprocess.run([PROBE2, "li", ind, outd, file, tag], cwd=SCENES_DIR, check=True)
        if "ref_hsppm.npz" in which:
            file, tag, photons, radius = HSPPM
            np.array([photons, radius], "<f4").tofile(os.path.join(ind, f"hsppm_{tag}.f32"))
            for it in range(2):
                O.photon_draws(SEED, 2, it * photons, photons, 21).tofile(os.path.join(ind, f"draws_{tag}_hsppm_it{it}.f32"))
            subprocess.run([PROBE2, "hsppm", ind, outd, file, tag], cwd=SCENES_DIR, check=True)
        arrays = {k: v for k, v in collect(outd).items() if not k.endswith("_draws")}
    arrays["paths_patch"] = arrays["li_patch"] = arrays["hsppm_patch"] = np.array(DRAW_PATCH)
    return arrays


def renders(scene, out_dir):
    """SppmTracer::render as upstream runs it (its own threads and unseeded random numbers), several runs per scene:
    the mean image before gamma and its per-pixel variance from run to run.  Not reproducible run by run."""
    r = RENDER_SHAPE
    with tempfile.TemporaryDirectory() as tmp:
        ind, outd = os.path.join(tmp, "in"), os.path.join(tmp, "out")
        os.makedirs(ind)
        os.makedirs(outd)
        write_camera(scene, ind)
        for file, tag, csize in RENDERS:
            subprocess.run([PROBE2, "render", ind, outd, file, tag, str(r["W"]), str(r["H"]), str(r["recurse"]), str(r["gsize"]),
                            str(csize), str(r["runs"]), str(r["threads"])], cwd=SCENES_DIR, check=True)
            a = collect(outd)
            imgs = a[f"render_{tag}_images"].astype(np.float64)
            clipped = np.clip(imgs, 0.0, 1.0)       # what the image keeps of a pixel (src/tracer/TkSppmTracer.cpp:525-527)
            sel = {"mean": imgs.mean(0).astype(np.float32), "var": imgs.var(0, ddof=1).astype(np.float32),
                   "mean_clipped": clipped.mean(0).astype(np.float32), "var_clipped": clipped.var(0, ddof=1).astype(np.float32),
                   "counters": a[f"render_{tag}_counters"], "meta": a[f"render_{tag}_meta"],
                   "scene": np.array(file),
                   "note": np.array("meta: W H recurse gsize csize runs threads; counters per run: shootGlobalPhotons "
                                    "shootCausticPhotons numRecurse; images are Ld/numRecurse + caustic + global before clamp and gamma")}
            out = os.path.join(out_dir, f"ref_image_{tag}.npz")
            np.savez_compressed(out, **sel)
            print(os.path.basename(out), os.path.getsize(out) // 1024, "KiB")


def main(argv=None):
    import argparse
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=GOLDEN, help="where the fixtures are written (default: tests/golden)")
    ap.add_argument("--only", default="", help="comma-separated fixture names; default: every reproducible one")
    ap.add_argument("--renders", action="store_true", help="also (or, with --only none, only) the whole-render image fixtures: minutes")
    args = ap.parse_args(argv)
    for exe in (PROBE, PROBE2):
        if not os.path.exists(exe):
            raise SystemExit(f"{exe} missing: run `bash oracle/build_ref.sh` first")
    from pathtracer_amd.scene import Scene, asset_scene
    scene = Scene.load(asset_scene("cornell_sphere.scene"))
    want = [w for w in args.only.split(",") if w] or list(GROUPS) + list(GROUPS2)
    os.makedirs(args.out, exist_ok=True)
    arrays = {}
    if any(w in GROUPS for w in want):
        arrays.update(first_prober(scene, GOLDEN))
    if any(w in GROUPS2 for w in want):
        arrays.update(second_prober(scene, want))
    for out, prefixes in list(GROUPS.items()) + list(GROUPS2.items()):
        if out not in want:
            continue
        sel = {k: v for k, v in arrays.items() if any(k.startswith(p) for p in prefixes)}
        np.savez_compressed(os.path.join(args.out, out), **sel)
        print(out, len(sel), "arrays", os.path.getsize(os.path.join(args.out, out)) // 1024, "KiB")
    if args.renders:
        renders(scene, args.out)


if __name__ == "__main__":
    main()
