"""One SPPM render over `world` processes, one GPU each (torch.distributed).

The reference is a single process (shared-memory threads, src/Threads.cpp);
the path shards along its two natural seams:

  photon phase : rank r traces the light paths whose Halton index is r modulo
                 world and fills a quota of size/world photons per map;
  exchange     : the per-rank photon arrays are all-gathered (RCCL over xGMI
                 with the "nccl" backend) and the shoot counters all-reduced,
                 so every rank holds both complete maps and builds both trees;
  camera phase : rank r owns the pixel rows y with (y // 8) % world == r and
                 keeps their SPPMPixel state for the whole render;
  end          : the pixel planes are gathered to rank 0, which resolves.

The exchange logic is independent of what computes the phases: `HipBackend`
drives libtkhip on a GPU; the CPU tests plug the oracle in to check the
sharding arithmetic with the gloo backend.
"""
import ctypes as C

import numpy as np
import torch
import torch.distributed as dist

from . import _abi
from ._abi import check

ROW_BLOCK = 8


def row_owner(y, world):
    return (y // ROW_BLOCK) % world


def quota(size, rank, world):
    return size // world + (1 if rank < size % world else 0)


class HipBackend:
    """Phases of an iteration on one GPU through the C ABI; buffers are torch CUDA tensors."""

    def __init__(self, scene, param, max_depth, rr, seed, width, height, rank, world, device):
        from .tracer import SppmTracer
        self.tracer = SppmTracer(1, max_depth, 0, rr, param, device=device, seed=seed, rank=rank, world=world)
        self.tracer.setScene(scene)
        self.tracer.setCamera(width, height)
        self.tracer.startRaytracing()
        self.param, self.rank, self.world = param, rank, world
        self.width, self.height = width, height
        self.dev = torch.device("cuda", device)
        self.lib = self.tracer.ctx.lib
        self.h = self.tracer.ctx.h
        self.quota = [quota(param.globalSize, rank, world), quota(param.causticSize, rank, world)]
        self.max_quota = [quota(param.globalSize, 0, world), quota(param.causticSize, 0, world)]
        self.num_recurse = self.tracer.num_recurse

    def iteration_single(self, it, stats=None):
        """world == 1: the library overlaps the photon passes, the tree builds and the camera paths."""
        lib = self.lib
        check(lib, lib.tkhip_sppm_iteration(self.h, it, C.byref(stats) if stats is not None else None),
              "tkhip_sppm_iteration")

    def photon_pass(self, it, stats=None):
        self.tracer.photon_pass(it, stats)

    def export_local_into(self, which, pos, power, direction):
        """This rank's photons of one map into caller-owned device arrays ((cap, 3) float32 twice, (cap,) int16,
        rows past the photon count left as they are); returns (count, light paths shot)."""
        n, shot = C.c_int64(), C.c_int64()
        torch.cuda.current_stream(self.dev).synchronize()   # torch's work on these buffers, not the lanes of the library
        check(self.lib, self.lib.tkhip_sppm_export_local_photons(self.h, which, pos.data_ptr(), power.data_ptr(),
                                                                 direction.data_ptr(), pos.shape[0], C.byref(n), C.byref(shot)),
              "tkhip_sppm_export_local_photons")
        return n.value, shot.value

    def export_local(self, which):
        cap = max(self.max_quota[which], 1)
        pos = torch.zeros((cap, 3), dtype=torch.float32, device=self.dev)
        power = torch.zeros((cap, 3), dtype=torch.float32, device=self.dev)
        direction = torch.zeros((cap,), dtype=torch.int16, device=self.dev)
        n, shot = self.export_local_into(which, pos, power, direction)
        return pos, power, direction, n, shot

    def set_map(self, which, pos, power, direction, n, shot_total):
        torch.cuda.current_stream(self.dev).synchronize()   # torch's work on these buffers, not the lanes of the library
        check(self.lib, self.lib.tkhip_sppm_set_map_photons(self.h, which, pos.data_ptr(), power.data_ptr(),
                                                            direction.data_ptr(), n, shot_total),
              "tkhip_sppm_set_map_photons")

    # the exchange through one buffer per rank, laid out and unpacked by the library on the device
    def exchange_words(self):
        w = C.c_int64()
        check(self.lib, self.lib.tkhip_sppm_exchange_words(self.h, C.byref(w)), "tkhip_sppm_exchange_words")
        return w.value

    def export_exchange(self, buf):
        """This rank's header and photons of both maps into `buf` (int32, exchange_words() long) on torch's current
        stream; nothing is waited for."""
        stream = torch.cuda.current_stream(self.dev).cuda_stream
        check(self.lib, self.lib.tkhip_sppm_export_exchange(self.h, buf.data_ptr(), stream), "tkhip_sppm_export_exchange")

    def import_exchange(self, gathered):
        """All ranks' buffers back to back (as the all-gather leaves them on torch's current stream) become the two
        maps; the counts stay on the device (build_maps settles them)."""
        stream = torch.cuda.current_stream(self.dev).cuda_stream
        check(self.lib, self.lib.tkhip_sppm_import_exchange(self.h, gathered.data_ptr(), stream), "tkhip_sppm_import_exchange")

    def build_maps(self, stats=None):
        self.tracer.build_maps(stats)

    def photon_begin(self, it):
        check(self.lib, self.lib.tkhip_sppm_photon_begin(self.h, it), "tkhip_sppm_photon_begin")

    def camera_begin(self, it):
        check(self.lib, self.lib.tkhip_sppm_camera_begin(self.h, it), "tkhip_sppm_camera_begin")

    def camera_pass(self, it, stats=None, next_ahead=False):
        self.tracer.camera_pass(it, stats, next_ahead)

    def export_planes(self):
        planes = torch.empty((13, self.height, self.width), dtype=torch.float32, device=self.dev)
        check(self.lib, self.lib.tkhip_sppm_export_planes(self.h, planes.data_ptr()), "tkhip_sppm_export_planes")
        return planes

    def import_planes(self, planes):
        planes = planes.contiguous()
        torch.cuda.current_stream(self.dev).synchronize()   # torch's work on these buffers, not the lanes of the library
        check(self.lib, self.lib.tkhip_sppm_import_planes(self.h, planes.data_ptr()), "tkhip_sppm_import_planes")

    def resolve(self):
        return self.tracer.resolve(self.num_recurse)

    def close(self):
        self.tracer.close()


def _on_host(t):
    """gloo (used to rehearse the multi-rank path where RCCL cannot run, e.g. several ranks on one GPU) gathers
    host tensors only; RCCL takes the device tensors as they are."""
    return t.is_cuda and dist.get_backend() == "gloo"


def _all_gather(t, world):
    if _on_host(t):
        h = t.cpu()
        parts = [torch.empty_like(h) for _ in range(world)]
        dist.all_gather(parts, h)
        return [p.to(t.device) for p in parts]
    parts = [torch.empty_like(t) for _ in range(world)]
    dist.all_gather(parts, t.contiguous())
    return parts


def _all_gather_flat(t, world):
    """All ranks' equally long 1-D tensors back to back in one tensor on t's device."""
    if _on_host(t):
        h = t.cpu()
        out = torch.empty((world * h.numel(),), dtype=h.dtype)
        dist.all_gather_into_tensor(out, h)
        return out.to(t.device)
    out = torch.empty((world * t.numel(),), dtype=t.dtype, device=t.device)
    dist.all_gather_into_tensor(out, t)
    return out


def _exchange_layout(caps):
    """Word offsets inside one rank's exchange buffer (int32 words): 8 header words, then per map the
    positions (3 words per photon), the powers (3) and the direction pairs (2 bytes per photon, padded to
    whole words).  Returns ([(positions, powers, directions, direction words, capacity)] per map, total)."""
    off, lay = 8, []
    for cap in caps:
        dwords = (cap + 1) // 2
        lay.append((off, off + 3 * cap, off + 6 * cap, dwords, cap))
        off += 6 * cap + dwords
    return lay, off


def exchange_photons(backend, world):
    """All-gather of both maps' photons and sum of their shot counters, as ONE collective per iteration: each
    rank sends one flat int32 buffer (neither RCCL nor gloo gathers int16), a header (photon counts and shot
    counters of both maps) followed by the bit-cast arrays of both maps.  A backend with `export_local_into`
    writes its photons straight into that buffer.  Returns [(n_total, shot_total)] per map."""
    if world == 1:
        return [None, None]   # a single rank installs its own photons at the end of its photon pass
    if hasattr(backend, "export_exchange"):
        # The library lays the buffer out, fills it and unpacks the gathered buffers on the device: between the
        # photon pass and the tree builds the host reads nothing and waits for nothing (with RCCL; gloo, the
        # rehearsal backend, moves the buffers through host memory).  Counts and shot totals are settled by
        # build_maps, so there are no totals to return here.
        send = getattr(backend, "_exchange_send", None)
        if send is None:
            send = backend._exchange_send = torch.empty((backend.exchange_words(),), dtype=torch.int32, device=backend.dev)
        backend.export_exchange(send)
        gathered = _all_gather_flat(send, world)
        backend.import_exchange(gathered)
        backend._exchange_gathered = gathered      # stays alive until the next exchange replaces it
        return [None, None]
    direct = hasattr(backend, "export_local_into")
    if direct:
        caps = [max(q, 1) for q in backend.max_quota]
        dev = backend.dev
        parts = None
    else:
        parts = [backend.export_local(which) for which in (0, 1)]
        caps = [p[0].shape[0] for p in parts]
        dev = parts[0][0].device
    lay, total = _exchange_layout(caps)
    flat = getattr(backend, "_exchange_buffer", None)   # reused every iteration: what is read of it is rewritten
    if flat is None or flat.numel() != total or flat.device != dev:
        flat = torch.zeros((total,), dtype=torch.int32, device=dev)
        try:
            backend._exchange_buffer = flat
        except AttributeError:
            pass
    head = []
    for which, (o_pos, o_pow, o_dir, dwords, cap) in enumerate(lay):
        pos = flat[o_pos:o_pos + 3 * cap].view(torch.float32).view(cap, 3)
        power = flat[o_pow:o_pow + 3 * cap].view(torch.float32).view(cap, 3)
        direction = flat[o_dir:o_dir + dwords].view(torch.int16)[:cap]
        if direct:
            n, shot = backend.export_local_into(which, pos, power, direction)
        else:
            pos.copy_(parts[which][0])
            power.copy_(parts[which][1])
            direction.copy_(parts[which][2])
            n, shot = parts[which][3], parts[which][4]
        head += [n, shot & 0x7FFFFFFF, shot >> 31]
    flat[:6] = torch.tensor(head, dtype=torch.int32)
    flats = _all_gather(flat, world)
    heads = torch.stack([t[:6] for t in flats]).cpu().tolist()
    totals = []
    for which, (o_pos, o_pow, o_dir, dwords, cap) in enumerate(lay):
        counts = [int(h[3 * which]) for h in heads]
        shot_total = sum(int(h[3 * which + 1]) + (int(h[3 * which + 2]) << 31) for h in heads)
        n_total = sum(counts)
        pos = torch.cat([t[o_pos:o_pos + 3 * c] for t, c in zip(flats, counts)]).view(torch.float32).view(-1, 3)
        power = torch.cat([t[o_pow:o_pow + 3 * c] for t, c in zip(flats, counts)]).view(torch.float32).view(-1, 3)
        direction = torch.cat([t[o_dir:o_dir + dwords].view(torch.int16)[:c] for t, c in zip(flats, counts)])
        backend.set_map(which, pos, power, direction, n_total, shot_total)
        totals.append((n_total, shot_total))
    return totals


def iteration(backend, it, world, stats=None, begin_next=False):
    """photon pass -> exchange -> tree builds -> camera pass (gathers), for every rank in lockstep.
    begin_next 1 (or True): the light paths of iteration it + 1 are enqueued right after the exchange (they
    depend on nothing this iteration still computes) and run under its tree builds and gathers; 2: its camera
    paths are started as well, before this iteration's gathers (their queries take the pixels' radii again
    when they are gathered).  The next call must then be for it + 1."""
    if world == 1 and hasattr(backend, "iteration_single") and not begin_next and getattr(backend, "_lights_begun", None) != it:
        backend.iteration_single(it, stats)
        return [None, None]
    ahead = int(begin_next) if hasattr(backend, "camera_begin") else 0
    if hasattr(backend, "camera_begin"):
        # camera paths do not read the maps: trace them beside the photon phase, behind the light paths
        # in the GPU's queues (the exchange waits for those)
        if getattr(backend, "_lights_begun", None) != it:
            backend.photon_begin(it)
        backend._lights_begun = None
        backend.camera_begin(it)   # nothing to do when the last call began them ahead
    backend.photon_pass(it, stats)
    totals = exchange_photons(backend, world)
    if ahead:
        backend.photon_begin(it + 1)
        backend._lights_begun = it + 1
    backend.build_maps(stats)
    if ahead >= 2:
        # ... and the camera paths of iteration it + 1 as well, before this iteration's gathers
        backend.camera_pass(it, stats, next_ahead=True)
    else:
        backend.camera_pass(it, stats)
    return totals


def gather_planes(backend, rank, world, height):
    """Merges the row-sharded SPPMPixel planes on rank 0 (other ranks return None)."""
    planes = backend.export_planes()
    if world == 1:
        return planes
    if _on_host(planes):
        host = planes.cpu()
        parts = [torch.empty_like(host) for _ in range(world)] if rank == 0 else None
        dist.gather(host, parts, dst=0)
        parts = [p.to(planes.device) for p in parts] if rank == 0 else None
    else:
        parts = [torch.empty_like(planes) for _ in range(world)] if rank == 0 else None
        dist.gather(planes, parts, dst=0)
    if rank != 0:
        return None
    owner = torch.tensor([row_owner(y, world) for y in range(height)], device=planes.device)
    merged = planes.clone()
    for r in range(1, world):
        rows = owner == r
        merged[:, rows, :] = parts[r][:, rows, :]
    return merged


def init_process_group_from_env(backend="nccl"):
    """RANK / WORLD_SIZE / MASTER_* come from torch.distributed.run."""
    if not dist.is_initialized():
        dist.init_process_group(backend=backend)
    return dist.get_rank(), dist.get_world_size()
