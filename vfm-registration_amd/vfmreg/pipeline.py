"""Device-resident registration pipeline: the hot path of
RegistrationNode.ransac_registration('vfm') (registration_node.py:273-328) as HIP kernels with no
host synchronisation:

    normalise + MFMA fragment tiles   (VoxelHashMap.cpp:469-482; int8 image for the int8 coarse pass, fp16 for the fp16 one)
 -> top-1 inner-product search        (VoxelHashMap.cpp:486-495)
 -> cosine threshold + compaction      (VoxelHashMap.cpp:501-511, 587-600)
 -> correspondence RANSAC + Kabsch     (registration_node.py:319-327)

Buffers are allocated once for fixed (N, M, D); ``register`` only enqueues work.

``overlap_ransac=True`` turns the chain into a pipeline over independent scene pairs:
stage 1 (caller's stream) = operand preparation (HBM-bound) and the MFMA coarse pass of pair i+1;
stage 2 (``solve_streams`` side streams, round-robin) = candidate selection / rescan / fp32 refinement, exact fp64
re-decision, threshold / compaction and RANSAC of pairs i, i-1, ...  Events order the hand-offs and ``solve_streams + 1``
complete buffer sets (prepared operands, search workspace, results) rotate, so results of a pair stay valid until
``solve_streams + 1`` further pairs have been enqueued.  Coarse passes never overlap each other; the inputs of a pair must
stay untouched until its ``done`` event.
``overlap_prepare=True`` moves the operand preparation to a stream of its own, beside the coarse pass of the previous
pair.  With the fp16 pass it cost the coarse kernel more than it hid (round 2, first half: 360 vs 365 registrations/s);
with the int8 pass, whose kernel is half as long, prepare stream + two solve streams is the best arrangement measured
(604-665 registrations/s against 558-574 for one solve stream with prepare on the caller's stream) and is what bench.py runs.

``coarse``: which coarse pass -- "int8-half" = the half-width pass of the gated family (VFM_RECORDS_HALF: int8 MFMA over the
first d / 2 columns, the other half bounded by Cauchy-Schwarz against the gate; needs the gate), "int8" / "int8-top2" = the
gated family of include/vfmreg.h with best-score / packed top-2 records (queries that provably miss ``min_cosine`` stay unresolved: idx -1, sim -2.0; correspondences and pose are
unaffected), "fp16" = the fp16 pass (VFM_RECORDS_F16: every query resolved), "auto" (default) = the half-width pass if a probe of it (first registration,
then every ``REPROBE``) and every search after that report at most ``HALF_LIMIT`` surviving chunks per query, else best-score
records until a search reports more than
``RESCAN_LIMIT`` rescanned chunks per query (duplicate-rich maps), then top-2 records, then -- above ``TOP2_LIMIT`` -- the fp16
pass, with a probe one step back every ``REPROBE`` registrations.  ``gate=False`` keeps the int8 pass but resolves every query.
"""
from __future__ import annotations

from typing import Optional

import numpy as np
import torch

from . import _lib, ops


class _ResultSet:
    def __init__(self, n: int, dev, qprep_bytes: int, bprep_bytes: int, sws_bytes: int):
        u8 = torch.uint8
        self.qprep = torch.empty(qprep_bytes, dtype=u8, device=dev)
        self.bprep = torch.empty(bprep_bytes, dtype=u8, device=dev)
        self.sws = torch.empty(sws_bytes, dtype=u8, device=dev)
        self.map_key: Optional[int] = None
        self.idx = torch.empty(n, dtype=torch.int64, device=dev)
        self.sim = torch.empty(n, dtype=torch.float32, device=dev)
        self.keep = torch.empty(n, dtype=torch.int64, device=dev)
        self.count = torch.zeros(1, dtype=torch.int64, device=dev)
        self.corres = torch.empty((n, 2), dtype=torch.int32, device=dev)
        self.T = torch.empty((4, 4), dtype=torch.float64, device=dev)
        self.fitness = torch.empty(1, dtype=torch.float64, device=dev)
        self.rmse = torch.empty(1, dtype=torch.float64, device=dev)
        self.best_hyp = torch.empty(1, dtype=torch.int32, device=dev)
        self.mask = torch.empty(n, dtype=torch.uint8, device=dev)
        self.done: Optional[torch.cuda.Event] = None  # RANSAC of the pair that used this set finished


# Side streams are shared by every pipeline of a process (per device): HIP spreads the streams a process creates over a few
# hardware queues (4 by default), and streams on one queue run one after the other.  Which queues a NEW stream lands on depends
# on how many were created before it: the same pipeline, built again later in the process, ran at 1110, 1270 or 1375
# registrations/s depending on whether a solve stream had come to share the coarse stream's queue (tools/queue_probe.py,
# tools/queue_probe_trace.sh).  The first streams a process creates get queues of their own; they are kept and reused.
# Consequences a caller should know: the streams are never released; two pipelines of one process -- or pipelines driven from
# different threads -- serialise their side stages on these streams (results stay correct: every hand-off is ordered by events).
# ``RegistrationPipeline(private_streams=True)`` gives a pipeline streams of its own and is the choice for concurrent pipelines;
# the queue placement it then gets is the HIP runtime's (an observed heuristic, not a contract).
_SIDE_STREAMS = {}
# priority of the preparation stream / the solve streams when they are first created (0 = default, -1 = high); A/B runs set these
# before the first pipeline of the process is built (tools/ab_priority.py)
PREP_STREAM_PRIORITY = 0
SOLVE_STREAM_PRIORITY = 0


# A HIP stream is bound to one of the process's hardware queues (four by default, round robin) when it is first USED, not when it is created.
# The side streams are therefore used once, in a fixed order, the moment they are created: preparation, solve 0, solve 1 take the three queues
# behind the default stream's whatever the program does between building a pipeline and its first registration, and the feature stream of
# the C3 pipelines can be placed relative to them (_feature_stream).  Measured neutral for the C2 pipeline, plain and under
# torch.distributed.run (tools/ab_queue_touch.sh).  TOUCH_STREAMS_AT_CREATION = False restores the lazy binding (A/B).
TOUCH_STREAMS_AT_CREATION = True


def _touch(stream: torch.cuda.Stream, dev: torch.device) -> None:
    if TOUCH_STREAMS_AT_CREATION:
        with torch.cuda.stream(stream):
            torch.zeros(1, device=dev)


def _side_streams(dev: torch.device, n_solve: int):
    key = (dev.index if dev.index is not None else torch.cuda.current_device())
    prep, solve = _SIDE_STREAMS.get(key, (None, []))
    if prep is None:
        prep = torch.cuda.Stream(device=dev, priority=PREP_STREAM_PRIORITY)
        _touch(prep, dev)
    while len(solve) < n_solve:
        solve.append(torch.cuda.Stream(device=dev, priority=SOLVE_STREAM_PRIORITY))
        _touch(solve[-1], dev)
    _SIDE_STREAMS[key] = (prep, solve)
    return prep, solve[:n_solve]


_FEATURE_STREAMS = {}
_PLACEHOLDER_STREAMS = []
E2E_SOLVE_STREAMS = 2    # solve streams of EndToEndPipeline's registration pipeline (tools/time_c3_group.py VFM_E2E_SOLVE: A/B)
FEATURE_QUEUE_SKIP = 2   # placeholder streams used in front of the feature stream (tools/time_c3_group.py VFM_FEATURE_SKIP: A/B)


def _feature_stream(dev: torch.device, priority: int):
    """The feature stage's stream, one per (device, priority) and process for the same reason as ``_side_streams``: which hardware
    queue a new stream lands on depends on how many the process created before, and the same EndToEndPipeline ran at 400, 470 or
    520 registrations/s depending on it (tools/time_c3_group.py, profiles/r04_time_c3_group.txt)."""
    key = (dev.index if dev.index is not None else torch.cuda.current_device(), int(priority))
    if key not in _FEATURE_STREAMS:
        # The feature stream is the fifth stream of a process whose pipeline has the default stream, the preparation stream and two solve
        # streams on the four hardware queues: bound next in the round robin it would land on the DEFAULT stream's queue, and the ViT's 63
        # launches would queue behind the 0.7 ms coarse kernels they are meant to run beside (453 instead of 517 - 524 registrations/s from
        # uint8 images, 565 instead of 627 in groups of four: tools/time_c3_group.py with VFM_PREAMBLE / VFM_FEATURE_SKIP).  Two placeholder
        # streams are used first, so that it shares the first solve stream's queue instead (three: the second's, 500 / 617).
        for _ in range(FEATURE_QUEUE_SKIP if TOUCH_STREAMS_AT_CREATION else 0):
            ph = torch.cuda.Stream(device=dev)
            with torch.cuda.stream(ph):
                torch.zeros(1, device=dev)
            _PLACEHOLDER_STREAMS.append(ph)
        fs = torch.cuda.Stream(device=dev, priority=int(priority))
        with torch.cuda.stream(fs):
            torch.zeros(1, device=dev)
        _FEATURE_STREAMS[key] = fs
    return _FEATURE_STREAMS[key]


class RegistrationPipeline:
    def __init__(self, n: int, m: int, d: int = 384, n_iter: int = 50000, min_cosine: float = 0.8,
                 max_corr_dist: float = 10000.0, seed: int = 42, device="cuda", overlap_ransac: bool = False,
                 overlap_prepare: bool = False, solve_streams: int = 1, gate: bool = True, coarse: str = "auto",
                 half_fused: Optional[bool] = None, prep_schedule: Optional[int] = None, private_streams: bool = False,
                 config: Optional["_lib.Config"] = None):
        lib = _lib.load()
        # kernel policy of THIS pipeline's library calls (round 6: a caller-owned vfm_config_t, include/vfmreg.h): bound to the calling
        # thread for the length of every register() / prepare_map(); None = whatever the calling thread has bound (factory settings
        # if nothing).  Two pipelines with different configs may run from two threads at once.
        self.config = config
        self.n, self.m, self.d = n, m, d
        self.n_iter, self.min_cosine, self.max_corr_dist, self.seed = n_iter, min_cosine, max_corr_dist, seed
        self.device = torch.device(device)
        dev = self.device
        u8 = torch.uint8
        self.overlap = bool(overlap_ransac)
        self.overlap_prepare = bool(overlap_prepare)
        self.gate = bool(gate)
        # which coarse pass: "int8" = the gated family of include/vfmreg.h with best-score records (the cheapest kernel; every
        # candidate chunk of a resolved query is rescanned), "int8-top2" = the same with packed top-2 records (+ ~0.15 ms of
        # kernel at C2; a chunk with one row inside the bounds costs one fp32 row instead of a 48 KB rescan), "fp16" = the
        # ungated family, "auto" = chosen from the searches' own feedback (_poll_feedback)
        if coarse not in ("auto", "int8-half", "int8", "int8-top2", "mx6", "mx6-top2", "mx6-pilot", "mx6-fused", "mx6-half", "fp16"):
            raise ValueError("coarse must be 'auto', 'int8-half', 'int8', 'int8-top2', 'mx6', 'mx6-top2', 'mx6-half' or 'fp16'")
        if coarse in ("int8-half", "mx6-half", "mx6-fused") and not gate:
            raise ValueError("the half-width pass (and the fused full-width one) need the gate")
        self.coarse = coarse
        self.use_i8 = coarse != "fp16"
        self.top2 = coarse == "int8-top2"   # int8 pass with packed top-2 records (VFM_RECORDS_TOP2)
        # the full-width coarse pass in microscaled fp6 (VFM_RECORDS_MX6: twice the int8 instruction's rate, ~3x wider bounds;
        # the operands are prepared with VFM_PREPARE_MX6); where the library has no kernel for it, best-score records
        self.mx6 = coarse in ("mx6", "mx6-top2", "mx6-pilot", "mx6-fused")   # "mx6-top2": the same pass with packed top-2 records (VFM_RECORDS_MX6_TOP2)
        # "mx6-fused": VFM_RECORDS_MX6_FUSED (round 5) -- the full-width fp6 pass lists the (query, chunk) pairs that reach the gate in its
        # own epilogue and writes no records (no record array, no selection sweep); prunes where few rows reach the gate (D.2), falls
        # through to the guard's full-width int8 pass where many do -- a pinned mode for such data, not one `auto` picks
        self.mx6_fused = coarse == "mx6-fused"
        self.mx6_top2 = coarse == "mx6-top2"
        # "mx6-pilot": VFM_RECORDS_MX6_PILOT -- one chunk per query rescanned exactly in front of the selection (about half the candidate
        # chunks where a query has many near neighbours); `auto` turns it on above PILOT_UP rescanned chunks per query
        self.mx6_pilot = coarse == "mx6-pilot"
        # half-width pass (VFM_RECORDS_HALF): where the library has no kernel for it the call behaves as best-score records
        # ... and where almost every chunk survives its bound (descriptors that are all alike) it is slower than the full-width
        # modes -- bounded by the library's device-side guard (csrc/match_finish.hip: above 48 survivors per query the search falls
        # through to one full-width pass with the gate as hit test: 6.8 ms per C2-size registration against 1.9; before the
        # guard: 171 ms): "auto" therefore PROBES it (vfm_match_search_probe_half: its coarse pass + a count of the survivors,
        # +0.7 ms once) on the first registration and at every re-probe interval, and switches to it only on a good count
        self.half = coarse in ("int8-half", "mx6-half")
        # the half-width pass in fp6 (VFM_RECORDS_MX6_HALF): the same bound on the scaled MFMA; operands prepared with VFM_PREPARE_MX6
        self.mx6_half = coarse == "mx6-half"
        # "auto": where the fp6 kernel exists the half-width pass it settles on is the fp6 one (0.40 against 0.60 ms of coarse
        # kernel at C2 size, 1500-1600 against 1300-1360 registrations/s; the probe itself runs on the int8 half-width image, which
        # every preparation writes)
        self._mx6_half_ok = coarse == "auto" and d in (256, 384, 512, 768) and n > 2048
        # ... and where the half-width bound does not prune but best-score int8 records rescan only a few chunks per query (maps of
        # distinct places), the full-width pass moves to fp6 too: its bounds are ~3x wider, so it is tried below MX6_UP rescanned
        # chunks per query and left again above MX6_DOWN (tools/ab_mx6_bench.py: lifted descriptors of independent scenes, 6.2 int8 /
        # 21 fp6 rescans per query: 732 / 793 against 651 / 721 registrations/s; with a common component, 12.5 / 45: 590 against 630)
        self._mx6_ok = self._mx6_half_ok and d in (256, 384)   # (the full-width fp6 kernel: two query sets of d / 64 k-steps in registers)
        self._mx6_tried = False
        self._probe_due = coarse == "auto" and self.gate
        # what `auto` may use, as decided above: register(reuse_map=True) turns the fp6 passes off for as long as a reused map is
        # searched and restores these afterwards (it used to clear them for the rest of the pipeline's life)
        self._mx6_allowed = (self._mx6_ok, self._mx6_half_ok)
        self._fp6_off_by_reuse = False
        self._map_prepared = False    # prepare_map() was called: the buffer sets hold a map without the fp6 image
        # which form of the half-width pass: with the selection fused into the coarse kernel (VFM_RECORDS_HALF_FUSED = 4: no
        # records, no selection kernel) a serial registration is 1.5 % faster (1017 vs 1002 registrations/s), but in the
        # overlapped pipeline it is 1-2 % slower (1363 vs 1378; 1190 vs 1211 in 20-step runs): the selection kernel ran on a side
        # stream for free, the fused form adds a memset and the bin atomics to the stream everything waits for
        # (``half_fused``: None = that rule; True / False force VFM_RECORDS_HALF_FUSED / VFM_RECORDS_HALF -- A/B runs)
        self._half_kind = (3 if self.overlap else 4) if half_fused is None else (4 if half_fused else 3)
        # the fp6 half-width pass: VFM_RECORDS_MX6_HALF_FUSED = 8 (round 4: the coarse kernel lists its survivors itself -- in the LDS,
        # flushed once per workgroup by plain stores -- and writes no records: no 122 MB record array, no selection sweep) unless
        # ``half_fused=False`` asks for VFM_RECORDS_MX6_HALF = 7 (records + match_select_half_kernel; A/B runs)
        self._mx6_half_kind = 7 if half_fused is False else 8
        self.last_rescans: Optional[int] = None
        self.last_probe: Optional[int] = None
        self._since_switch = 0
        self._pending = []  # (event, pinned int32[1]) of gated searches whose rescan count is on its way to the host
        self._slots = []    # pinned slots ready for reuse
        # solve_streams = K: the solve stages of K consecutive pairs may run beside each other (and beside the coarse pass
        # of a later pair) on K side streams, with K + 1 buffer sets
        self.n_solve = max(1, int(solve_streams)) if self.overlap else 0
        sizes = (lib.vfm_match_prepared_bytes(n, d), lib.vfm_match_prepared_bytes(m, d),
                 lib.vfm_match_search_workspace_bytes(n, m, d))
        self.sets = [_ResultSet(n, dev, *sizes) for _ in range(self.n_solve + 1)]
        # (stream priorities were measured with the half-width pass: the coarse stream on high priority 1040 vs 1220
        # registrations/s -- the side stages starve and the pipeline stalls on its own dependencies; the side streams on high
        # priority 1260 vs 1267: no difference)
        shared_prep, shared_solve = (None, []) if (private_streams or not self.overlap) else _side_streams(dev, self.n_solve)
        self.solve_streams = shared_solve if shared_solve else [torch.cuda.Stream(device=dev) for _ in range(self.n_solve)]
        self.rws_list = [torch.empty(lib.vfm_ransac_workspace_bytes(n, n_iter), dtype=u8, device=dev)
                         for _ in range(max(1, self.n_solve))]
        self.rws = self.rws_list[0]
        self.ransac_stream = self.solve_streams[0] if self.overlap else None
        # (tools/trace_pipe.sh: the preparation kernel needs whole compute units -- 1024 threads x 127 registers -- and so does a
        # coarse workgroup: neither starts while the solve stage's small kernels sit on every compute unit, so a cycle is coarse
        # kernel + ~0.2 ms of small kernels + preparation, whatever the streams allow.  Measured and dropped in round 3: a
        # high-priority preparation stream (1000 vs 1380/s), a high-priority coarse stream, enqueueing the solve stage of
        # registration i behind the preparation of i + 1 (the coarse kernel then waits for the same small kernels))
        self.prep_stream = (shared_prep if shared_prep is not None else torch.cuda.Stream(device=dev)) if self.overlap else None
        self._step = 0
        self._prep_schedule = prep_schedule   # None: the rule in register(); 1 / 2 = VFM_PREPARE_PERSISTENT / _INTERLEAVED (A/B runs)

    def prepare_map(self, b_desc: torch.Tensor) -> None:
        """IndexFlatIP.add: normalise + convert the map once (it is immutable per scene); every buffer
        set gets its own copy so that ``register(..., reuse_map=True)`` never re-prepares."""
        lib = _lib.load()
        ops._chk(b_desc, torch.float32, "b_desc")
        if b_desc.shape != (self.m, self.d):
            raise ValueError("Invalid shape")
        # a map prepared once carries the fp16 and int8 images, not the fp6 one (vfm_match_prepare): the fp6 kinds are out
        if self.coarse in ("mx6", "mx6-top2", "mx6-pilot", "mx6-fused", "mx6-half"):
            raise ValueError("the fp6 modes prepare map and scan together in every registration: no prepare_map()")
        self._mx6_ok = self._mx6_half_ok = False
        self.mx6 = self.mx6_half = False
        self._map_prepared = True
        main = torch.cuda.current_stream()
        for r in self.sets:
            if r.done is not None:
                main.wait_event(r.done)
            _lib.check(lib.vfm_match_prepare(b_desc.data_ptr(), self.m, self.d, r.bprep.data_ptr(), main.cuda_stream),
                       "prepare(map)")
            r.map_key = b_desc.data_ptr()

    # Feedback thresholds, in rescanned chunks per query over all queries of the scan (tools/time_neardup.py, C2 size, ms per
    # registration with best-score vs top-2 records): 0.5 per query: 1.28 vs 1.53; 1.2: 1.33 vs 1.51; 6.2: 1.50 vs 1.73; 7.7: 1.81
    # vs 1.85; 12.4: 2.01 vs 1.94; 31.6: 2.57 vs 2.28; 99: 10.8 vs 3.4 (the chunk-major rescan, match_rescan_chunk_kernel, moved
    # the crossover from ~2.5 to ~10); top-2 records -> fp16 pass above TOP2_LIMIT (whole-chunk rescans + 1/32 per single row;
    # never reached on the maps measured: the fp16 pass takes 4.2 ms where top-2 records take 3.4)
    # half-width pass -> best-score records above HALF_LIMIT surviving chunks per query (tools/time_neardup.py, ms per
    # registration, half-width vs best-score records: D.2 descriptors, 0.5 survivors per query -- the planted matches and nothing
    # else: 0.87 vs 1.48; lifted, independent views, 20 per query: 1.23 vs 1.60; lifted, shared scene, 96: 8.3 vs 1.8; descriptors
    # that are all alike (C3): every chunk, 195 vs 2.4)
    # (round 3, with the chunk-major rescan on the matrix cores and the side streams on queues of their own -- the earlier
    # figures carried a +-15 % scatter from the hardware queues -- tools/ab_mx6_bench.py, 20 / 200 steps: 6.2 rescanned chunks per
    # query: best-score 651 / 721 vs top-2 593 / 655 registrations/s; 12.5: 630 / 683 vs 599 / 649; the crossover of round 2 is at
    # ~30 (profiles/r03_neardup.json: 9.4 per query 3.16 vs 3.23 ms))
    # (end of round 3, after the finish stage's rework -- candidates and hits staged in the LDS, the bin counters one per
    # 128-byte line, the chunk-major rescan fed by an LDS-DMA ring: a candidate chunk costs a third of what it did -- best-score
    # records win on every map of profiles/r03_neardup.json, in fp6 where the kernel exists: every point seen by 200 clouds,
    # 99 rescanned chunks per query: fp6 2.48 / int8 2.67 / top-2 3.17 ms; by 50 clouds: 1.55 / 1.70 / 1.89; lifted + common
    # component, 12.5 int8 / 46 fp6 rescans: 654 / 633 / 585 registrations/s.  The limits now sit beyond the measured range.)
    HALF_LIMIT = 24.0
    RESCAN_LIMIT = 128.0
    MX6_UP = 128.0
    MX6_DOWN = 400.0
    PILOT_UP = 1.0e9   # rescanned chunks per query above which `auto` adds the pilot rescan to the full-width fp6 pass (A/B: tools/ab_pilot.py)
    TOP2_LIMIT = 40
    REPROBE = 256       # registrations before one step back towards the cheaper kernel is probed

    def _poll_feedback(self) -> None:
        """Non-blocking: consume the rescan counts that have arrived and pick the coarse pass of the next registrations."""
        while self._pending and self._pending[0][0].query():
            _, slot, records = self._pending.pop(0)
            if records == "probe":  # survivors the half-width pass would leave (vfm_match_search_probe_half)
                self.last_probe = int(slot.item())
                self._slots.append(slot)
                if self.coarse == "auto" and self.use_i8 and not self.half and self.last_probe <= self.HALF_LIMIT * self.n:
                    self.half, self.top2, self.mx6 = True, False, False
                    self.mx6_half = self._mx6_half_ok
                    self._since_switch = 0
                continue
            self.last_rescans = int(slot.item())
            self._slots.append(slot)
            if self.coarse != "auto" or not self.use_i8 or records != self._records():
                continue  # feedback of a mode that has been left already
            if self.half:
                if self.last_rescans > self.HALF_LIMIT * self.n:
                    # the fp6 image's wider bounds leave more survivors than the probe -- taken on the int8 image -- counted: where
                    # the probe itself was inside the limit the half-width pass stays, on the int8 image (lifted descriptors of
                    # independent views, 20 / ~40 survivors per query: 0.98 ms against 1.05 for fp6 best-score records)
                    if self.mx6_half and self.last_probe is not None and self.last_probe <= self.HALF_LIMIT * self.n:
                        self.mx6_half = False
                    else:
                        self.half = False
                    self._since_switch = 0
            elif self.mx6:   # (only "auto" gets here: a pinned mode returned above)
                if self.last_rescans > self.MX6_DOWN * self.n:
                    self.mx6, self._mx6_tried = False, True
                    self._since_switch = 0
                elif not self.mx6_pilot and self.last_rescans > self.PILOT_UP * self.n:
                    self.mx6_pilot = True
                    self._since_switch = 0
            elif not self.top2 and self.last_rescans > self.RESCAN_LIMIT * self.n:
                self.top2 = True
                self._since_switch = 0
            elif not self.top2 and self._mx6_ok and not self._mx6_tried and self.last_rescans <= self.MX6_UP * self.n:
                self.mx6 = True
                self._since_switch = 0
            elif self.top2 and self.last_rescans > self.TOP2_LIMIT * self.n:
                self.use_i8 = False
                self._since_switch = 0

    def _records(self) -> int:
        if self.mx6:
            if self.mx6_fused:
                return 10                      # VFM_RECORDS_MX6_FUSED
            return 6 if self.mx6_top2 else (9 if self.mx6_pilot else 5)   # VFM_RECORDS_MX6_TOP2 / VFM_RECORDS_MX6_PILOT / VFM_RECORDS_MX6
        if self.half and self.mx6_half:
            return self._mx6_half_kind         # VFM_RECORDS_MX6_HALF_FUSED (VFM_RECORDS_MX6_HALF on request)
        return self._half_kind if self.half else (1 if self.top2 else 0)   # 4 = VFM_RECORDS_HALF_FUSED (falls back to 3 / 0 inside the library)

    def synchronize(self) -> None:
        """Make the caller's current stream wait for every RANSAC issued on the side stream."""
        for s in self.solve_streams:
            torch.cuda.current_stream().wait_stream(s)

    def register(self, q_desc: torch.Tensor, q_xyz: torch.Tensor, b_desc: torch.Tensor, b_xyz: torch.Tensor,
                 reuse_map: bool = False, want_mask: bool = True, inputs_ready: Optional[torch.cuda.Event] = None):
        if self.config is None:
            return self._register(q_desc, q_xyz, b_desc, b_xyz, reuse_map, want_mask, inputs_ready)
        with _lib.using(self.config):
            return self._register(q_desc, q_xyz, b_desc, b_xyz, reuse_map, want_mask, inputs_ready)

    def _register(self, q_desc: torch.Tensor, q_xyz: torch.Tensor, b_desc: torch.Tensor, b_xyz: torch.Tensor,
                  reuse_map: bool = False, want_mask: bool = True, inputs_ready: Optional[torch.cuda.Event] = None):
        """Enqueue one registration.  ``inputs_ready``: an event after which the four input tensors are complete (inputs
        produced on a stream other than the caller's current one).  Every mode waits for it before the first kernel that
        reads an input; with ``overlap_prepare`` it also spares the prepare stage from queueing behind the coarse pass of the
        previous pair on the caller's stream (without the event, it conservatively does)."""
        lib = _lib.load()
        # descriptor rows: float32 (the reference's layout behind VoxelHashMap.cpp:469-482) or float16 storage (round 5, BASELINE.json
        # configs[4]: half the bytes of a resident map; every element is widened to fp32 as the kernels load it -- the result is that
        # of the widened rows, include/vfmreg.h VFM_ROWS_F16).  fp16 rows go through the gated int8 / fp6 passes only.
        f16q, f16b = q_desc.dtype == torch.float16, b_desc.dtype == torch.float16
        ops._chk(q_desc, torch.float16 if f16q else torch.float32, "q_desc")
        ops._chk(b_desc, torch.float16 if f16b else torch.float32, "b_desc")
        ops._chk(q_xyz, torch.float64, "q_xyz")
        ops._chk(b_xyz, torch.float64, "b_xyz")
        if q_desc.shape != (self.n, self.d) or b_desc.shape != (self.m, self.d):
            raise ValueError("Invalid shape")
        if (f16q or f16b) and (reuse_map or self.coarse == "fp16" or self._map_prepared):
            raise ValueError("float16 descriptor rows: the int8 / fp6 passes with map and scan prepared together (no reuse_map, no fp16 pass)")
        if reuse_map:
            # a reused map is prepared once (vfm_match_prepare2 / vfm_match_prepare below): it carries the fp16 and int8 images, not
            # the fp6 one -- its err6 would be read as infinite and an fp6 search would prune nothing.  Same rule as prepare_map().
            if self.coarse in ("mx6", "mx6-top2", "mx6-pilot", "mx6-fused", "mx6-half"):
                raise ValueError("the fp6 modes prepare map and scan together in every registration: no reuse_map")
            self._mx6_ok = self._mx6_half_ok = False
            self.mx6 = self.mx6_half = False
            self._fp6_off_by_reuse = True
        elif self._fp6_off_by_reuse and not self._map_prepared:
            # back to registrations that prepare map and scan together: the fp6 passes are available again, and the policy decides
            # anew from the next searches' feedback (half-width probe first)
            self._mx6_ok, self._mx6_half_ok = self._mx6_allowed
            self._fp6_off_by_reuse = False
            self._mx6_tried = False
            self._probe_due = self.coarse == "auto" and self.gate and not self.half
            if self.coarse == "auto" and self.half:
                self.mx6_half = self._mx6_half_ok      # the half-width pass the probe had chosen, on the fp6 image again
        self._poll_feedback()
        if self.coarse == "auto":
            self._since_switch += 1
            if self._since_switch >= self.REPROBE and not self.half:
                # every REPROBE registrations one step back towards the cheaper kernel (fp16 -> top-2 records -> best-score
                # records); the feedback of that search decides whether it stays.  The half-width pass is probed, not tried.
                if not self.use_i8:
                    self.use_i8, self.top2 = True, True
                elif self.top2:
                    self.top2 = False
                self._mx6_tried = False
                self._probe_due = self.gate
                self._since_switch = 0
        if (f16q or f16b) and not self.use_i8:
            # the policy is decided ABOVE (feedback polled, re-probe applied) and only then read.  `auto` may have
            # left the int8 passes for the fp16 one (above TOP2_LIMIT rescans per query: duplicate-rich maps) -- the fp16 pass and its
            # finish stage read float32 rows (vfm_match_prepare2, vfm_match_search_finish_gated_r with records 2): float16 storage must
            # never reach them.  Such rows stay on the int8 pass with packed top-2 records, the widest gated mode that widens on load.
            self.use_i8, self.top2 = True, True
            self.half = self.mx6 = False
        i8, records = self.use_i8, self._records()
        assert i8 or not (f16q or f16b)
        r = self.sets[self._step % len(self.sets)]
        solve = self.solve_streams[self._step % self.n_solve] if self.overlap else None
        rws = self.rws_list[self._step % self.n_solve] if self.overlap else self.rws
        self._step += 1
        main = torch.cuda.current_stream()
        st = main.cuda_stream
        pst = st
        if not (self.overlap and self.overlap_prepare) and inputs_ready is not None:
            main.wait_event(inputs_ready)               # inputs produced on another stream (was ignored here)
        if self.overlap and not self.overlap_prepare:
            if r.done is not None and not r.done.query():
                main.wait_event(r.done)                 # the solve stage that last read this set has finished
        elif self.overlap:
            # stage 0 on its own stream: it may run beside the coarse pass of the previous pair
            if inputs_ready is not None:
                self.prep_stream.wait_event(inputs_ready)
            else:
                self.prep_stream.wait_stream(main)      # inputs produced on the caller's stream are ready
            if r.done is not None:
                self.prep_stream.wait_event(r.done)     # the solve stage that last read this set has finished
            pst = self.prep_stream.cuda_stream
        if not (reuse_map and r.map_key == b_desc.data_ptr()):
            # (a map that will be reused keeps both images: the coarse pass may change between registrations)
            if i8 and not reuse_map:
                # the preparation kernel's launch shape: persistent when it runs alone or beside the half-width coarse kernel
                # (which leaves registers free), short workgroups beside the full-width one (include/vfmreg.h)
                # (re-measured on the stable pipeline, 20 / 200 steps: fp6 half-width 1345-1371 / 1568 persistent against 1364-1387 / 1578
                # interleaved; int8 full width 701-704 / 779 against 705-707 / 788)
                schedule = 1 if (records in (3, 4) or not (self.overlap and self.overlap_prepare)) else 2
                if self._prep_schedule is not None:
                    schedule = int(self._prep_schedule)
                if records in (5, 6, 9, 10):
                    schedule |= 8   # VFM_PREPARE_MX6: the fp6 image as well
                elif records in (7, 8):
                    # VFM_PREPARE_MX6_HALF: the half-width pass reads the first d / 2 columns of the fp6 image -- only those are
                    # converted, and no int8 half-width image is written (the probe that needs it runs outside this mode)
                    schedule |= 8 | 16
                if f16q or f16b:
                    _lib.check(lib.vfm_match_prepare2_gated_t(b_desc.data_ptr(), int(f16b), self.m, r.bprep.data_ptr(), q_desc.data_ptr(), int(f16q),
                                                              self.n, r.qprep.data_ptr(), self.d, schedule, pst), "prepare(map + scan)")
                else:
                    _lib.check(lib.vfm_match_prepare2_gated_p(b_desc.data_ptr(), self.m, r.bprep.data_ptr(), q_desc.data_ptr(), self.n,
                                                              r.qprep.data_ptr(), self.d, schedule, pst), "prepare(map + scan)")
            else:
                _lib.check(lib.vfm_match_prepare2(b_desc.data_ptr(), self.m, r.bprep.data_ptr(), q_desc.data_ptr(), self.n,
                                                  r.qprep.data_ptr(), self.d, pst), "prepare(map + scan)")
            r.map_key = b_desc.data_ptr() if reuse_map else None
        else:
            _lib.check(lib.vfm_match_prepare(q_desc.data_ptr(), self.n, self.d, r.qprep.data_ptr(), pst), "prepare(scan)")
        if self.overlap and pst != st:
            main.wait_stream(self.prep_stream)
        gate = float(np.nextafter(np.float32(self.min_cosine), np.float32(-np.inf))) if self.gate else float("-inf")
        if i8 and self._probe_due and not self.half and len(self._pending) < 8:
            self._probe_due = False
            slot = self._slots.pop() if self._slots else torch.zeros(1, dtype=torch.int32).pin_memory()
            _lib.check(lib.vfm_match_search_probe_half(r.qprep.data_ptr(), self.n, r.bprep.data_ptr(), self.m, self.d, r.sws.data_ptr(),
                                                       r.sws.numel(), gate, slot.data_ptr(), st), "probe(half)")
            ev = torch.cuda.Event()
            ev.record(main)
            self._pending.append((ev, slot, "probe"))
        if i8:
            _lib.check(lib.vfm_match_search_coarse_gated_g(r.qprep.data_ptr(), self.n, r.bprep.data_ptr(), self.m, self.d,
                                                           r.sws.data_ptr(), r.sws.numel(), records, gate, st), "search(coarse)")
        else:
            # (VFM_RECORDS_F16 = 2: the fp16 pass explicitly -- the ungated calls route large searches to the int8 pass)
            _lib.check(lib.vfm_match_search_coarse_gated_r(r.qprep.data_ptr(), self.n, r.bprep.data_ptr(), self.m, self.d,
                                                           r.sws.data_ptr(), r.sws.numel(), 2, st), "search(coarse)")
        rst = st
        if self.overlap:  # hand over to stage 2
            ev = torch.cuda.Event()
            ev.record(main)
            solve.wait_event(ev)
            rst = solve.cuda_stream
        # only matches with cosine >= min_cosine are kept below: queries that provably cannot reach it stay unresolved (gate)
        if i8 and (f16q or f16b):
            _lib.check(lib.vfm_match_search_finish_gated_t(q_desc.data_ptr(), int(f16q), r.qprep.data_ptr(), self.n, b_desc.data_ptr(), int(f16b),
                                                           r.bprep.data_ptr(), self.m, self.d, r.idx.data_ptr(), r.sim.data_ptr(),
                                                           r.sws.data_ptr(), r.sws.numel(), gate, records, rst), "search(finish)")
        if i8 and not (f16q or f16b):
            _lib.check(lib.vfm_match_search_finish_gated_r(q_desc.data_ptr(), r.qprep.data_ptr(), self.n, b_desc.data_ptr(),
                                                           r.bprep.data_ptr(), self.m, self.d, r.idx.data_ptr(), r.sim.data_ptr(),
                                                           r.sws.data_ptr(), r.sws.numel(), gate, records, rst), "search(finish)")
        if i8:
            if self.coarse == "auto" and len(self._pending) < 8:  # feedback: candidate chunks this search rescans
                slot = self._slots.pop() if self._slots else torch.zeros(1, dtype=torch.int32).pin_memory()
                _lib.check(lib.vfm_match_search_rescans_async(r.sws.data_ptr(), self.n, self.m, slot.data_ptr(), rst), "rescans")
                ev = torch.cuda.Event()
                ev.record(solve if self.overlap else main)
                self._pending.append((ev, slot, records))
        else:
            _lib.check(lib.vfm_match_search_finish_gated_r(q_desc.data_ptr(), r.qprep.data_ptr(), self.n, b_desc.data_ptr(),
                                                           r.bprep.data_ptr(), self.m, self.d, r.idx.data_ptr(), r.sim.data_ptr(),
                                                           r.sws.data_ptr(), r.sws.numel(), float("-inf"), 2, rst), "search(finish)")
        _lib.check(lib.vfm_threshold_compact(r.sim.data_ptr(), r.idx.data_ptr(), self.n, float(self.min_cosine),
                                             r.keep.data_ptr(), r.count.data_ptr(), r.corres.data_ptr(),
                                             None, None, None, None, rst), "threshold_compact")
        _lib.check(lib.vfm_ransac_corr(q_xyz.data_ptr(), b_xyz.data_ptr(), r.corres.data_ptr(), r.count.data_ptr(),
                                       self.n, float(self.max_corr_dist), int(self.n_iter), int(self.seed),
                                       r.T.data_ptr(), r.fitness.data_ptr(), r.rmse.data_ptr(),
                                       r.mask.data_ptr() if want_mask else None, r.best_hyp.data_ptr(),
                                       rws.data_ptr(), rws.numel(), rst), "ransac")
        if self.overlap:
            r.done = torch.cuda.Event()
            r.done.record(solve)
        return dict(T=r.T, fitness=r.fitness, rmse=r.rmse, best_hyp=r.best_hyp, mask=r.mask, idx=r.idx, sim=r.sim,
                    keep=r.keep, count=r.count, corres=r.corres, done=r.done,
                    result_stream=solve if self.overlap else main)


def cu_mask_words(ncu: int, offset: int = 0, total: int = 256) -> list:
    """The 32-bit words of a compute-unit mask with bits offset .. offset + ncu - 1 set, wrapping around ``total`` units --
    ceil(total / 32) words (parts whose unit count is not a multiple of 32: 304, 228, 110 ...)."""
    if not (0 < ncu <= total) or offset < 0:
        raise ValueError("Invalid compute-unit range")
    words = [0] * ((total + 31) // 32)
    for i in range(offset, offset + ncu):
        j = i % total                   # wrap around the chip, THEN split into word / bit
        words[j // 32] |= 1 << (j % 32)
    return words


def masked_stream(ncu: int, offset: int = 0, total: int = 256) -> "torch.cuda.Stream":
    """A HIP stream whose kernels may use ``ncu`` of the ``total`` compute units (hipExtStreamCreateWithCUMask, mask bits
    offset .. offset + ncu - 1), as a torch stream."""
    import ctypes as C
    hip = C.CDLL("libamdhip64.so")   # (the runtime torch has loaded)
    w = cu_mask_words(ncu, offset, total)
    words = (C.c_uint32 * len(w))(*w)
    hip.hipExtStreamCreateWithCUMask.argtypes = [C.POINTER(C.c_void_p), C.c_uint32, C.POINTER(C.c_uint32)]
    hip.hipExtStreamCreateWithCUMask.restype = C.c_int
    st = C.c_void_p()
    rc = hip.hipExtStreamCreateWithCUMask(C.byref(st), len(w), words)
    if rc != 0:
        raise RuntimeError(f"hipExtStreamCreateWithCUMask failed ({rc})")
    return torch.cuda.ExternalStream(st.value)


class EndToEndPipeline:
    """Config C3 as a pipeline (round 4): surround images + scan points in, pose out, over a sequence of
    independent pairs -- create_descriptors (prepare_scenes.py:50-107) feeding ransac_registration('vfm') (registration_node.py:273-328).

    The feature stage of pair i + 1 -- ViT-S/14 on its cameras (63 dependent launches of 5 - 15 us: latency-bound, a few percent of
    the chip) and the fused projection + lifting -- runs on a stream of its own beside the registration of pair i, whose coarse pass
    is one fat MFMA kernel; ``RegistrationPipeline`` overlaps the solve stage of pair i - 1 as before.  ``depth`` buffer sets
    (images, patch grids, lifted descriptors) rotate; a set is reused only when the registration that read it has finished (its
    ``done`` event).  Results are those of ``model.forward`` + ``LiftPlan`` + ``RegistrationPipeline.register`` run one after the
    other (tests/test_gpu_e2e.py)."""

    def __init__(self, model, cams: list, n: int, m: int, n_iter: int = 50000, min_cosine: float = 0.8, max_corr_dist: float = 10000.0,
                 seed: int = 42, depth: int = 4, coarse: str = "auto", device="cuda", feature_cus: int = 0, feature_priority: int = 0,
                 group: int = 1, group_depth: int = 3):
        """``model``: vit.ViTS14 for the rig's image size; ``cams``: the rig, one dict per camera in priority order with the
        projection parameters of ``ops.LiftPlan`` (mode, mats, fc, subsample, win, H, W, rot_mode) -- image and grid pointers are
        the pipeline's own.  ``group`` > 1 allocates ``group_depth`` buffer sets for ``submit_group``: the cameras of up to ``group``
        pairs go through the ViT in ONE call (6 images are 63 launches of 5 - 15 us each, bounded by their boundaries; 24 images cost
        2.3x of that, not 4x -- the batch path prepare_scenes.create_descriptors_batch uses offline, here inside the pipeline)."""
        self.model, self.n = model, n
        self.device = torch.device(device)
        d = model.dim
        B, H, W = len(cams), model.img_h, model.img_w
        self.reg = RegistrationPipeline(n, m, d, n_iter=n_iter, min_cosine=min_cosine, max_corr_dist=max_corr_dist, seed=seed,
                                        device=device, overlap_ransac=True, overlap_prepare=True, solve_streams=E2E_SOLVE_STREAMS, coarse=coarse)
        self.depth = max(int(depth), len(self.reg.sets) + 1)
        # ``feature_cus`` > 0: the feature stage on a stream restricted to that many compute units and the registration's main stream
        # (operand preparation hand-off + coarse pass) on the others.  A coarse workgroup owns its compute unit (8 waves x ~200
        # registers), so beside an unrestricted coarse kernel each of the ViT's 63 dependent launches waits for coarse workgroups to
        # end before it gets a compute unit at all; with a few units of its own the latency-bound forward runs at its own pace.
        self.feature_cus = int(feature_cus)
        if self.feature_cus > 0:
            ncu = torch.cuda.get_device_properties(self.device).multi_processor_count
            self.feat_stream = masked_stream(self.feature_cus, 0, ncu)
            self.reg_stream = masked_stream(ncu - self.feature_cus, self.feature_cus, ncu)
        else:
            # (``feature_priority`` < 0: a high-priority stream -- its workgroups are placed first whenever a compute unit has room)
            self.feat_stream = _feature_stream(self.device, feature_priority)
            self.reg_stream = None
        self.sets = []
        for _ in range(self.depth):
            imgs = torch.empty((B, H, W, 3), dtype=torch.uint8, device=self.device)
            grids = torch.empty((B, 16, model.patch_w, d), dtype=torch.float32, device=self.device)
            desc = torch.empty((n, d), dtype=torch.float32, device=self.device)
            filled = torch.zeros(n, dtype=torch.uint8, device=self.device)
            pcl = torch.empty((4, n), dtype=torch.float64, device=self.device)
            plan = ops.LiftPlan([dict(c, proj_image=(imgs[k] if c.get("needs_image") else None), grid=grids[k], Hup=H, Wup=W,
                                      raw_image=imgs[k]) for k, c in enumerate(cams)], d)
            self.sets.append(dict(imgs=imgs, grids=grids, desc=desc, filled=filled, pcl=pcl, plan=plan, done=None, ready=None))
        self._step = 0
        self.group = max(int(group), 1)
        self.gsets = []
        if self.group > 1:
            G = self.group
            for _ in range(max(int(group_depth), 2)):
                imgs = torch.empty((G * B, H, W, 3), dtype=torch.uint8, device=self.device)
                grids = torch.empty((G * B, 16, model.patch_w, d), dtype=torch.float32, device=self.device)
                desc = torch.empty((G, n, d), dtype=torch.float32, device=self.device)
                filled = torch.zeros((G, n), dtype=torch.uint8, device=self.device)
                pcl = torch.empty((G, 4, n), dtype=torch.float64, device=self.device)
                plans = [ops.LiftPlan([dict(c, proj_image=(imgs[g * B + k] if c.get("needs_image") else None), grid=grids[g * B + k],
                                            Hup=H, Wup=W, raw_image=imgs[g * B + k]) for k, c in enumerate(cams)], d) for g in range(G)]
                self.gsets.append(dict(imgs=imgs, grids=grids, desc=desc, filled=filled, pcl=pcl, plans=plans, done=[]))
        self._gstep = 0
        self._cams = B

    def submit(self, images: torch.Tensor, pcl4xn: torch.Tensor, q_xyz: torch.Tensor, b_desc: torch.Tensor, b_xyz: torch.Tensor,
               inputs_ready: Optional[torch.cuda.Event] = None, want_mask: bool = True):
        """Enqueue one pair: ``images`` [cameras, H, W, 3] uint8, ``pcl4xn`` [4, n] fp64 homogeneous scan points (PS:69), ``q_xyz``
        [n, 3] fp64 the same points for the solve, the map's descriptors and points.  Returns ``RegistrationPipeline.register``'s
        dict + ``desc`` (the lifted descriptors of this pair's buffer set, valid until ``depth`` further pairs were submitted)."""
        s = self.sets[self._step % self.depth]
        self._step += 1
        fs = self.feat_stream
        main = torch.cuda.current_stream()
        if inputs_ready is not None:
            fs.wait_event(inputs_ready)
        else:
            fs.wait_stream(main)
        if s["done"] is not None:
            fs.wait_event(s["done"])          # the registration that read this set's descriptors has finished
        with torch.cuda.stream(fs):
            s["imgs"].copy_(images, non_blocking=True)
            s["pcl"].copy_(pcl4xn, non_blocking=True)
            self.model.forward(s["imgs"], out=s["grids"])
            s["plan"](s["pcl"], s["desc"], s["filled"])
            s["ready"] = torch.cuda.Event()
            s["ready"].record(fs)
        if self.reg_stream is not None:
            self.reg_stream.wait_stream(main)
            with torch.cuda.stream(self.reg_stream):
                out = self.reg.register(s["desc"], q_xyz, b_desc, b_xyz, want_mask=want_mask, inputs_ready=s["ready"])
        else:
            out = self.reg.register(s["desc"], q_xyz, b_desc, b_xyz, want_mask=want_mask, inputs_ready=s["ready"])
        s["done"] = out["done"]
        out = dict(out)
        out["desc"] = s["desc"]
        return out

    def submit_group(self, pairs: list, inputs_ready: Optional[torch.cuda.Event] = None, want_mask: bool = True, on_result=None) -> list:
        """Enqueue 1 ... ``group`` pairs whose feature stages share one ViT call.  ``pairs``: a list of ``(images, pcl4xn, q_xyz,
        b_desc, b_xyz)`` as ``submit`` takes them.  The cameras of all pairs are copied into one [pairs x cameras, H, W, 3] buffer
        and go through ``model.forward`` once (the batch kernels are bit-identical to the one-scan kernels: tests/test_gpu_vit.py),
        then every pair is lifted from its own slice of the patch grids and registered as ``submit`` would, in order.  A
        registration's outputs live in ``RegistrationPipeline``'s rotating buffer sets, of which there are fewer than pairs in a
        group: ``on_result(k, out)`` is called right after pair k's registration was enqueued -- snapshot what you need there, on
        ``out["result_stream"]``.  Returns the list of the ``out`` dicts (the last ``len(reg.sets)`` are still valid).  The feature
        stage of the next group runs beside the registrations of this one; a group's buffers are reused ``group_depth`` groups
        later, when all of its registrations are done."""
        G = len(pairs)
        if not 1 <= G <= self.group or not self.gsets:
            raise ValueError(f"submit_group takes 1 ... group = {self.group} pairs (EndToEndPipeline(group=...)); got {G}")
        B = self._cams
        gs = self.gsets[self._gstep % len(self.gsets)]
        self._gstep += 1
        fs = self.feat_stream
        main = torch.cuda.current_stream()
        if inputs_ready is not None:
            fs.wait_event(inputs_ready)
        else:
            fs.wait_stream(main)
        for ev in gs["done"]:
            fs.wait_event(ev)                 # the registrations that read this set's descriptors have finished
        ready = []
        with torch.cuda.stream(fs):
            for k, p in enumerate(pairs):
                gs["imgs"][k * B:(k + 1) * B].copy_(p[0], non_blocking=True)
                gs["pcl"][k].copy_(p[1], non_blocking=True)
            self.model.forward(gs["imgs"][:G * B], out=gs["grids"][:G * B])
            for k in range(G):
                gs["plans"][k](gs["pcl"][k], gs["desc"][k], gs["filled"][k])
                ev = torch.cuda.Event()
                ev.record(fs)
                ready.append(ev)
        outs, done = [], []
        for k, p in enumerate(pairs):
            if self.reg_stream is not None:
                self.reg_stream.wait_stream(main)
                with torch.cuda.stream(self.reg_stream):
                    out = self.reg.register(gs["desc"][k], p[2], p[3], p[4], want_mask=want_mask, inputs_ready=ready[k])
            else:
                out = self.reg.register(gs["desc"][k], p[2], p[3], p[4], want_mask=want_mask, inputs_ready=ready[k])
            done.append(out["done"])
            out = dict(out)
            out["desc"] = gs["desc"][k]
            if on_result is not None:
                on_result(k, out)
            outs.append(out)
        gs["done"] = done
        return outs

    def synchronize(self) -> None:
        """Make the caller's current stream wait for everything submitted."""
        if self.reg_stream is not None:
            with torch.cuda.stream(self.reg_stream):
                self.reg.synchronize()
            torch.cuda.current_stream().wait_stream(self.reg_stream)
        else:
            self.reg.synchronize()
