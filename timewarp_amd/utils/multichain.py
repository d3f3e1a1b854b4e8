"""Several independent Markov chains evaluated in lock-step on one GPU (an extension -
the reference runs one chain, utils/evaluation_utils.py:517).

Each iteration draws S proposals for every one of the C chains and pushes all C*S rows through ONE flow reverse
pass, one energy evaluation and ONE flow forward pass (rows ordered sample-major, row = s*C + c, which is what the
reference's `[S, B]` reshape produces and what `tw_flow_sample_with_logp_multi` implements); the accept test runs
per chain (one workgroup per chain) and moves every chain's state on the device.  Per chain the arithmetic, the order of
its random draws and the emitted rows are exactly those of `sample_with_model(..., num_proposal_steps=S)` on that chain
alone, including the last-iteration clip `k = min(k, N - i)`; tests/test_mh_gpu.py checks this bit for bit.

With the HIP flow and the AMBER energy kernel an iteration of all C chains is ONE C-ABI call, `tw_mh_iteration_chains`
(r06; csrc/tw_mh_step.hip): four glue kernels around the two flow passes and one energy launch.  Without `noises` the
draws come from the call's own counter-based generator (Philox4x32-10 keyed (seed, chain, iteration), nothing drawn through
ATen); with `noises` (one source per chain: the oracle / trace-replay tests) the caller's draws are handed in.  Other
energy callables and `TW_MH_FUSED=0` take the op-by-op route below; the routes agree bit for bit on the same draws."""
from __future__ import annotations

from typing import List, Optional, Sequence

import ctypes as C
import os

import torch

from .. import _lib
from .evaluation_utils import (ChainStats, DeviceNoise, RecordingNoise, ReplayDraws, _deferred, _range_guarded,
                               check_symmetry_change, compute_kinetic_energy)


def _accept_chains(energy, p_xy, p_yx, u, y_c, y_v, x_c, x_v):
    S, Cn = energy.shape
    dev = energy.device
    ex = torch.empty((S, Cn), dtype=torch.float32, device=dev)
    p_acc = torch.empty((S, Cn), dtype=torch.float32, device=dev)
    acc = torch.empty((S, Cn), dtype=torch.uint8, device=dev)
    res = torch.empty((Cn, 4), dtype=torch.int32, device=dev)
    lib = _lib.load()
    with torch.cuda.device(dev):
        _lib.check(lib.tw_mh_accept_chains(energy.data_ptr(), p_xy.data_ptr(), p_yx.data_ptr(), u.data_ptr(), y_c.data_ptr(),
                                           y_v.data_ptr(), x_c.data_ptr(), x_v.data_ptr(), ex.data_ptr(), p_acc.data_ptr(),
                                           acc.data_ptr(), res.data_ptr(), S, Cn, x_c.shape[-2], _lib.stream_ptr(dev)),
                   "tw_mh_accept_chains")
    return ex, p_acc, acc, res


def draw_chains(model, device, seed: int, iteration: int, first_chain: int, S: int, n_chains: int, V: int):
    """The draws `tw_mh_iteration_chains` makes for (seed, iteration, chains first_chain .. first_chain + n_chains - 1), written
    out (tw_mh_draw_chains): latents z_coords, z_velocs [S, C, V, 3] (scaled by the model's prior), accept uniforms [S, C] and
    the resampled current velocities [C, V, 3].  For replays and tests; the iteration itself never materialises them."""
    dev = torch.device(device)
    f32 = torch.float32
    zc = torch.empty((S, n_chains, V, 3), dtype=f32, device=dev)
    zv = torch.empty_like(zc)
    u = torch.empty((S, n_chains), dtype=f32, device=dev)
    v = torch.empty((n_chains, V, 3), dtype=f32, device=dev)
    desc = model._desc(dev)
    raw, _ = model._weights(dev, model._path_for(V))
    draws = _lib.MHDraws(int(seed), int(iteration), int(first_chain), 1)
    with torch.cuda.device(dev):
        _lib.check(_lib.load().tw_mh_draw_chains(C.byref(desc), raw.data_ptr(), C.byref(draws), zc.data_ptr(), zv.data_ptr(),
                                                 u.data_ptr(), v.data_ptr(), S, n_chains, V, _lib.stream_ptr(dev)),
                   "tw_mh_draw_chains")
    return zc, zv, u, v


class MetropolisHastingsChains:
    """C chains of one molecule type (same atom types, masses and energy function), each with its own state and
    its own noise source."""

    KEYS = ("ind", "acc", "pxy", "pyx", "exp", "epot", "ekin", "dpot", "dkin")
    STAT_KEYS = KEYS[1:]   # the rows of the [8, S, C] statistics block (tw_mh_iteration_chains' out_stats)

    def __init__(self, batches: Sequence, model, device, energy_fn, masses, num_proposal_steps: int,
                 random_velocs: bool = False, resample_velocs: bool = False, reference_signs=None,
                 chirality_centers=None, noises: Optional[Sequence] = None, seed: Optional[int] = None, first_chain: int = 0):
        self.device = device = torch.device(device)
        self.C = C = len(batches)
        assert C >= 1 and all(b.atom_coords.size(0) == 1 for b in batches)
        self.model, self.energy_fn, self.S = model, energy_fn, int(num_proposal_steps)
        # draws: the caller's sources, or (None) the fused call's counter-based generator keyed (seed, first_chain + c, iteration)
        self.kernel_draws = noises is None
        self.noises = list(noises) if noises is not None else [DeviceNoise(device) for _ in range(C)]
        assert len(self.noises) == C
        # (no seed given: from the device's default generator, so torch.cuda.manual_seed decides the chains)
        self.seed = int(seed) if seed is not None else int(torch.randint(0, 2 ** 62, (1,), device=device).item())
        self.first_chain = int(first_chain)
        self.iteration = 0
        # split-fp16 range guard (modules/flow.py): keep the draws since the last read-back, see flush()
        self._guard = _range_guarded(model)
        if self._guard:
            self.noises = [RecordingNoise(n) for n in self.noises]
        f32 = torch.float32
        self.random_velocs, self.resample_velocs = random_velocs, resample_velocs
        self.x_coords = torch.cat([b.atom_coords.to(device, f32) for b in batches], dim=0).contiguous()
        self.x_velocs = torch.cat([n.randn_like(b.atom_coords.to(device, f32)) if random_velocs else b.atom_velocs.to(device, f32)
                                   for n, b in zip(self.noises, batches)], dim=0).contiguous()
        self.atom_types = torch.cat([b.atom_types.to(device) for b in batches], dim=0)
        self.masked = torch.cat([b.masked_elements.to(device) for b in batches], dim=0)
        self.masses = masses.to(device, f32)
        self.V = self.x_coords.shape[1]
        self.use_chirality = chirality_centers is not None and reference_signs is not None
        self.chirality_centers, self.reference_signs = chirality_centers, reference_signs
        self.kbT = energy_fn.kbT
        self.sgn = 1.0 if random_velocs else -1.0
        self.chain_c = [[self.x_coords[c:c + 1].clone()] for c in range(C)]
        self.chain_v = [[self.x_velocs[c:c + 1].clone()] for c in range(C)]
        self.rec = [{k: [] for k in self.KEYS} for _ in range(C)]
        self.emitted = [0] * C
        self.accepted = [0] * C
        self.proposals = 0
        self._pending = []
        self._inflight = None   # flush(lag=True): the window whose read-back is under way
        self._fused = self._fused_available()
        self.kernel_draws = self.kernel_draws and self._fused
        self._fconst = None

    # ---- one C-ABI call per iteration of all chains (tw_mh_iteration_chains) --------------------------------------------
    def _fused_available(self) -> bool:
        from ..energy import AmberPotentialEnergyTorch
        from ..modules.flow import ConditionalFlowDensityModel

        if os.environ.get("TW_MH_FUSED", "1") == "0":
            return False
        return (isinstance(self.model, ConditionalFlowDensityModel) and isinstance(self.energy_fn, AmberPotentialEnergyTorch)
                and not self.model.dims.ignore_cond_velocity and self.x_coords.is_cuda
                and self.energy_fn.tables.n_atoms == self.V)

    def _fused_constants(self):
        if self._fconst is None:
            dev, S, Cn, V = self.device, self.S, self.C, self.V
            lib = _lib.load()
            desc = self.model._desc(dev)
            opt = _lib.MHOptions()
            opt.random_velocs = int(self.random_velocs)
            keep = {"masses": self.masses.contiguous(), "types": self.atom_types.to(torch.int32).contiguous(),
                    "masked": self.masked.to(torch.uint8).contiguous()}
            opt.masses = keep["masses"].data_ptr()
            opt.kbT = float(self.kbT)
            opt.n_centres = 0
            if self.use_chirality:
                keep["centres"] = self.chirality_centers.to(dev, torch.int32).contiguous()
                keep["signs"] = self.reference_signs.to(dev, torch.float32).reshape(-1).contiguous()
                opt.n_centres = int(keep["centres"].shape[0])
                opt.centres, opt.reference_signs = keep["centres"].data_ptr(), keep["signs"].data_ptr()
            need = lib.tw_mh_iteration_chains_workspace_bytes(C.byref(desc), S, Cn, V)
            if need < 0:
                raise RuntimeError("tw_mh_iteration_chains_workspace_bytes failed: " + lib.tw_last_error().decode())
            self._fconst = dict(desc=desc, opt=opt, keep=keep, ws=torch.empty(int(need), dtype=torch.uint8, device=dev))
        return self._fconst

    def _step_fused(self) -> None:
        S, Cn, V, dev = self.S, self.C, self.V, self.device
        model = self.model
        fc = self._fused_constants()
        x_c, x_v = self.x_coords, self.x_velocs
        f32 = torch.float32
        zc = torch.empty(((S + 1) * Cn, V, 3), dtype=f32, device=dev)
        zv = torch.empty((S * Cn, V, 3), dtype=f32, device=dev)
        draws = None
        if self.kernel_draws:
            draws = _lib.MHDraws(self.seed, self.iteration, self.first_chain, int(self.random_velocs and self.resample_velocs))
            u = torch.empty((S, Cn), dtype=f32, device=dev)
        else:  # the callers' sources, chain by chain, in sample_with_model's draw order: velocities, latents, uniforms
            if self.random_velocs and self.resample_velocs:
                x_v = torch.cat([n.randn_like(x_v[c:c + 1]) for c, n in enumerate(self.noises)], dim=0).contiguous()
            sc = torch.exp(model.coords_prior_log_scale.detach()).to(dev)
            sv = torch.exp(model.velocs_prior_log_scale.detach()).to(dev)
            z3c, z3v = zc[: S * Cn].view(S, Cn, V, 3), zv.view(S, Cn, V, 3)
            for c, n in enumerate(self.noises):
                a, b = n.latents(S, 1, V, sc, sv)
                z3c[:, c].copy_(a.reshape(S, V, 3))
                z3v[:, c].copy_(b.reshape(S, V, 3))
            u = torch.stack([n.uniform(S).to(dev, f32) for n in self.noises], dim=1).contiguous()
        self.iteration += 1
        path = model._path_for(V)
        raw, packed = model._weights(dev, path)
        ff = self.energy_fn._device_ff(dev)
        cur_v = torch.empty_like(x_v)
        new_c, new_v = torch.empty_like(x_c), torch.empty_like(x_v)
        stats = torch.empty((8, S, Cn), dtype=f32, device=dev)
        acc = torch.empty((S, Cn), dtype=torch.uint8, device=dev)
        res = torch.empty((Cn, 4), dtype=torch.int32, device=dev)
        lib = _lib.load()
        with torch.cuda.device(dev):
            _lib.check(lib.tw_mh_iteration_chains(
                C.byref(fc["desc"]), raw.data_ptr(), _lib.ptr(packed), path, C.byref(ff.struct), C.byref(fc["opt"]),
                C.byref(draws) if draws is not None else None, fc["keep"]["types"].data_ptr(), fc["keep"]["masked"].data_ptr(), V,
                x_c.data_ptr(), x_v.data_ptr(), cur_v.data_ptr(), zc.data_ptr(), zv.data_ptr(), u.data_ptr(), new_c.data_ptr(),
                new_v.data_ptr(), stats.data_ptr(), acc.data_ptr(), res.data_ptr(), S, Cn, fc["ws"].data_ptr(), fc["ws"].numel(),
                _lib.stream_ptr(dev)), "tw_mh_iteration_chains")
        self.proposals += S * Cn
        self._pending.append((res, x_c, cur_v, new_c, new_v, acc, stats))
        self.x_coords, self.x_velocs = new_c, new_v

    def step_deferred(self) -> None:
        """One iteration of every chain, no host synchronisation."""
        S, C, V, dev = self.S, self.C, self.V, self.device
        model, kbT = self.model, self.kbT
        if not self._pending:
            self._pending_start = (self.x_coords, self.x_velocs, self.iteration)
            if self._guard:
                for n in self.noises:
                    n.mark()
        if self._fused:
            return self._step_fused()
        x_c, x_v = self.x_coords, self.x_velocs
        if self.random_velocs and self.resample_velocs:
            x_v = torch.cat([n.randn_like(x_v[c:c + 1]) for c, n in enumerate(self.noises)], dim=0)
        sc = torch.exp(model.coords_prior_log_scale.detach()).to(dev)
        sv = torch.exp(model.velocs_prior_log_scale.detach()).to(dev)
        lat = [n.latents(S, 1, V, sc, sv) for n in self.noises]
        z_c = torch.cat([l[0] for l in lat], dim=1).contiguous()
        z_v = torch.cat([l[1] for l in lat], dim=1).contiguous()
        with _deferred(model):  # the range flag is looked at in flush(), where the results are read back
            y_c, y_v, p_xy = model.conditional_sample_with_logp(
                atom_types=self.atom_types, x_coords=x_c, x_velocs=x_v, adj_list=None, edge_batch_idx=None,
                masked_elements=self.masked, num_samples=S, z_coords=z_c, z_velocs=z_v, allow_multi=True)
        rows_c, rows_v = y_c.reshape(S * C, V, 3), y_v.reshape(S * C, V, 3)
        e_pot_x = (self.energy_fn(x_c) / kbT).reshape(C)
        e_kin_x = compute_kinetic_energy(x_v, self.masses, random_velocs=self.random_velocs, kbT=kbT)
        e_kin_y = compute_kinetic_energy(rows_v, self.masses, random_velocs=self.random_velocs, kbT=kbT).reshape(S, C)
        e_pot_y = (self.energy_fn(rows_c) / kbT).reshape(S, C)
        if self.use_chirality:
            changed = check_symmetry_change(rows_c, self.chirality_centers, self.reference_signs).reshape(S, C)
            e_pot_y = torch.where(changed, e_pot_y + 2000, e_pot_y)
        e_kin = e_kin_y - e_kin_x[None]
        e_pot = e_pot_y - e_pot_x[None]
        energy = (e_pot + e_kin).contiguous()
        sgn = self.sgn
        with _deferred(model):
            p_yx = model.log_likelihood(
                atom_types=self.atom_types.repeat(S, 1), y_coords=x_c.repeat(S, 1, 1), y_velocs=(sgn * x_v).repeat(S, 1, 1),
                x_coords=rows_c, x_velocs=sgn * rows_v, adj_list=None, edge_batch_idx=None,
                masked_elements=self.masked.repeat(S, 1)).reshape(S, C).contiguous()
        p_xy = p_xy.reshape(S, C).contiguous()
        self.proposals += S * C
        u = torch.stack([n.uniform(S).to(dev, torch.float32) for n in self.noises], dim=1).contiguous()
        new_c, new_v = x_c.clone(), x_v.clone()
        ex, p_acc, acc, res = _accept_chains(energy, p_xy, p_yx, u, y_c.contiguous(), y_v.contiguous(), new_c, new_v)
        self._pending.append((res, x_c, x_v, new_c, new_v, acc,
                              torch.stack([p_acc, p_xy, p_yx, ex, e_pot_y, e_kin_y, e_pot, e_kin], dim=0)))   # STAT_KEYS order
        self.x_coords, self.x_velocs = new_c, new_v

    # ---- read-back and bookkeeping -----------------------------------------------------------------------------------------
    def _overflow_flag(self):
        """This model's range-guard word on the chains' device (None: never ran on a half-precision kernel there)."""
        if not self._guard:
            return None
        key = self.device.index if self.device.index is not None else torch.cuda.current_device()
        return getattr(self.model, "_range_flags", {}).get(key)

    def _snapshot(self):
        """Start the read-back of the parked window - 4 ints per (iteration, chain) and the range-guard word, into pinned
        memory behind the window's kernels - and hand the window over."""
        res = torch.stack([p[0] for p in self._pending])                       # [T, C, 4]
        host = torch.empty(res.shape, dtype=res.dtype, pin_memory=True)
        host.copy_(res, non_blocking=True)
        flag, flag_host = self._overflow_flag(), None
        if flag is not None:
            flag_host = torch.empty(1, dtype=torch.int32, pin_memory=True)
            flag_host.copy_(flag, non_blocking=True)
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream(self.device))
        win = dict(pending=self._pending, start=self._pending_start, host=host, flag=flag_host, ev=ev)
        self._pending = []
        return win

    def _overflowed(self, win) -> bool:
        # (`demoted`: someone else read - and cleared - the model's sticky flag while these iterations were parked)
        return self._guard and (bool(getattr(self.model, "demoted", False)) or (win["flag"] is not None and int(win["flag"][0]) != 0))

    def can_lag(self) -> bool:
        """flush(lag=True) is honoured when a replay needs no recorded draws: the kernel's own generator (a replay re-runs the
        same counters) or a model the range guard does not watch."""
        return self.kernel_draws or not self._guard

    def flush(self, num_samples: Optional[int] = None, lag: bool = False) -> None:
        """Bookkeeping of the parked iterations.  With `num_samples`, a chain that has emitted that many states
        ignores further iterations and its last counted iteration gets the reference's clip.
        lag=True (r06): only START the read-back of the window just queued and book the window BEFORE it, whose results
        arrived long ago - the host never waits for the device and the device never waits for the host's bookkeeping (the
        synchronous form leaves it idle for ~1 ms per read-back).  Counters (`accepted`, `emitted`, `chain_c` ...) then trail
        the queue by one window until a plain flush()."""
        if lag and self.can_lag():
            newer = self._snapshot() if self._pending else None
            older, self._inflight = self._inflight, newer
            if older is not None:
                self._finish(older, num_samples)
            return
        older, self._inflight = self._inflight, None
        if older is not None:
            self._finish(older, num_samples)
        if self._pending:
            self._finish(self._snapshot(), num_samples)

    def _finish(self, win, num_samples) -> None:
        """Book one window whose read-back was started; on a range-guard trip redo it - and everything queued behind it -
        on the exact-f32 kernels from its starting states with the same draws; the chains then continue there."""
        win["ev"].synchronize()
        if self._overflowed(win):
            torch.cuda.synchronize(self.device)
            behind = (len(self._inflight["pending"]) if self._inflight is not None else 0) + len(self._pending)
            self._inflight = None
            self.model.demote_to_f32()
            flag = self._overflow_flag()
            if flag is not None:
                flag.zero_()
            n_iter, recorders = len(win["pending"]) + behind, self.noises
            self.noises = [ReplayDraws(r.log) for r in recorders] if not self.kernel_draws else recorders
            self.x_coords, self.x_velocs, self.iteration = win["start"]   # (kernel draws: the same counters again)
            self.proposals -= n_iter * self.S * self.C
            self._pending, self._guard = [], False
            try:
                for _ in range(n_iter):
                    self.step_deferred()
            finally:
                self.noises = [getattr(r, "inner", r) for r in recorders]
            redo = self._snapshot()
            redo["ev"].synchronize()
            self._book(redo["host"].tolist(), redo["pending"], num_samples)
            return
        self._book(win["host"].tolist(), win["pending"], num_samples)

    def _book(self, results, pending, num_samples) -> None:
        """What a window of iterations emits, for all chains and iterations at once: the host decides the counts from the
        4 ints per (iteration, chain) and builds the gather indices; the device gathers the rows with one repeat_interleave
        per trajectory array and one index_select per statistics block - nothing here waits for the device (r05 did this slice
        by slice: ~12 tensor ops per chain and iteration, 10 ms of host time per 8 iterations of 32 chains, GPU idle)."""
        import numpy as np

        T, Cn, S, V, dev = len(pending), self.C, self.S, self.V, self.device
        cnt = np.zeros((Cn, T, 2), dtype=np.int64)     # [chain, iteration]: copies of the old state in front of the iteration's last row; 1 if it counts
        moved = np.zeros((Cn, T), dtype=bool)
        for t, per_chain in enumerate(results):
            for c, (k_true, any_acc, _, _) in enumerate(per_chain):
                if num_samples is not None and self.emitted[c] >= num_samples:
                    continue
                self.accepted[c] += int(any_acc)
                k = k_true if num_samples is None else min(k_true, num_samples - self.emitted[c])
                cnt[c, t, 0], cnt[c, t, 1], moved[c, t] = k, 1, bool(any_acc) and k == k_true
                self.emitted[c] += k + 1
        take = cnt.sum(axis=2)                          # [C, T]: rows 0 .. k of the iteration (or none)
        sizes = [int(x) for x in take.sum(axis=1)]
        total = sum(sizes)
        if total == 0:
            return
        up = lambda a: torch.from_numpy(np.ascontiguousarray(a)).pin_memory().to(dev, non_blocking=True)
        cnt_d, mv = up(cnt.reshape(-1)), up(moved)[:, :, None, None]
        for which, store in ((1, self.chain_c), (2, self.chain_v)):
            old = torch.stack([p[which] for p in pending], dim=1)              # [C, T, V, 3]
            new = torch.stack([p[which + 2] for p in pending], dim=1)
            src = torch.stack([old, torch.where(mv, new, old)], dim=2)         # [C, T, 2, V, 3]: the repeated row, the last row
            rows = torch.repeat_interleave(src.reshape(Cn * T * 2, V, 3), cnt_d, dim=0, output_size=total)
            for c, piece in enumerate(rows.split(sizes)):
                if sizes[c]:
                    store[c].append(piece)
        # statistics: element (t, s, c) of the stacked [T, S, C] blocks, chain-major then iteration then proposal
        flat_take = take.reshape(-1)
        within = np.arange(total) - np.repeat(np.cumsum(flat_take) - flat_take, flat_take)       # 0 .. k inside each (c, t)
        cc, tt = np.divmod(np.repeat(np.arange(Cn * T), flat_take), T)
        idx = up((tt * S + within) * Cn + cc)
        acc = torch.stack([p[5] for p in pending], dim=0).reshape(-1).index_select(0, idx).bool()              # [N]
        stats = torch.stack([p[6] for p in pending], dim=1).reshape(8, -1).index_select(1, idx)               # [8, N]
        for c, (a, st) in enumerate(zip(acc.split(sizes), stats.split(sizes, dim=1))):
            if sizes[c]:
                self.rec[c]["ind"].append(a)
                for name, row in zip(self.STAT_KEYS, st.unbind(0)):
                    self.rec[c][name].append(row)

    def results(self) -> List:
        """Per chain: (coords [1+n,V,3] numpy, velocs, accepted, ChainStats) as `sample_with_model` returns them."""
        self.flush()
        out = []
        for c in range(self.C):
            stats = ChainStats(*[torch.cat(self.rec[c][k], dim=0).cpu().numpy() for k in self.KEYS])
            out.append((torch.cat(self.chain_c[c], dim=0).cpu().numpy(), torch.cat(self.chain_v[c], dim=0).cpu().numpy(),
                        self.accepted[c], stats))
        return out


def sample_with_model_chains(batches: Sequence, model, device, openmm_potential_energy_torch, masses, num_samples: int,
                             num_proposal_steps: int, random_velocs: bool = False, resample_velocs: bool = False,
                             reference_signs=None, chirality_centers=None, noises: Optional[Sequence] = None,
                             sync_every: int = 8, seed: Optional[int] = None, first_chain: int = 0) -> List:
    """Run len(batches) chains (Metropolis-Hastings with `num_proposal_steps` parallel proposals each) until every
    one has emitted `num_samples` states; per chain the result equals `sample_with_model(batch, ..., num_samples,
    accept=True, num_proposal_steps=...)` driven by the same noise source."""
    chains = MetropolisHastingsChains(batches, model, device, openmm_potential_energy_torch, masses, num_proposal_steps,
                                      random_velocs=random_velocs, resample_velocs=resample_velocs,
                                      reference_signs=reference_signs, chirality_centers=chirality_centers, noises=noises,
                                      seed=seed, first_chain=first_chain)
    # lagged read-back (the host books window k - 1 while the device runs window k): the loop then notices one window late
    # that every chain is done, at most `sync_every` wasted iterations - taken when that is small against the run
    lag = num_samples >= 16 * max(1, sync_every)
    with torch.no_grad():
        while min(chains.emitted) < num_samples:
            for _ in range(max(1, sync_every)):
                chains.step_deferred()
            chains.flush(num_samples, lag=lag)
        chains.flush(num_samples)
    return chains.results()
