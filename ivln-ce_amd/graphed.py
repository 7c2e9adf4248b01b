"""hipGraph-replayed rollout step: mapper obs-transform + `policy.act` captured (through
torch.cuda.CUDAGraph - our kernels launch on torch's current stream, so stream capture records
them) and replayed per env step.  Inside the capture the three independent branches of the step run
on forked streams: instruction bi-LSTM || mapper -> semantic-map CNN || DD-PPO depth ResNet, joined
before the recurrent/attention head.  At 4-8 envs the step is ~170 launches of a few microseconds
each: replay removes the per-launch host cost and the fork overlaps the latency-bound branches.

Two graphs are captured with the recurrent state / previous action ping-ponging between two
buffer sets (graph 0: A -> B, graph 1: B -> A), so no state copies are needed between steps; the
only per-step copies are the observation tensors the step actually reads.
"""
import os
from typing import Dict

import torch

from . import mapping as _mapping
from . import ops

_streams = {}


def _stream(device, role, **kw):
    """One long-lived stream per (device, role): per-stream resources (split-K workspaces, 32 MB each) are keyed
    by the stream, so a fresh stream per capture - the eval loop re-captures whenever envs pause - would pin a
    new set every time."""
    key = (str(device), role)
    if key not in _streams:
        _streams[key] = torch.cuda.Stream(device, **kw)
    return _streams[key]


# `not_done_masks` resets the mapper (and the policy, unless `episode_not_done_masks` is given too: iterative
# evaluation keeps the maps for a whole tour while the policy state still resets per episode)
# `_u_sample` / `_u_beta` (+ the expert sensor named by `extra_keys`): the host-supplied uniforms of a SAMPLED step (DAgger
# collection): with them the action head draws and beta-mixes on the device (policy.ILPolicy._act), so that the sampled
# step is a pure function of its inputs and replays like the deterministic one
_STEP_KEYS = ("depth", "semantic12", "rgb", "instruction", "world_robot_pose", "world_robot_orientation",
              "not_done_masks", "episode_not_done_masks", "_u_sample", "_u_beta")


def _policy_masks(batch):
    return batch.get("episode_not_done_masks", batch["not_done_masks"])


class _CutCapture:
    """Stream capture of a SEQUENCE of hipGraphs that share one memory pool: `cut()` ends the graph being captured and begins
    the next one on the same stream, so that a long dependent chain (RedNet -> mapper -> map CNN) can be replayed in pieces
    with an event recorded between two of them - the point another stream's graph waits for.  Replaying the pieces back to
    back on one stream is the uncut graph (same launches, same order, same buffers)."""

    def __init__(self, stream, pool=None):
        self.stream, self.pool, self.graphs = stream, pool, []

    def _begin(self):
        g = torch.cuda.CUDAGraph()
        if self.pool is None:
            g.capture_begin()
        else:
            g.capture_begin(pool=self.pool)
        self.graphs.append(g)

    def cut(self):
        self.graphs[-1].capture_end()
        self.pool = self.graphs[-1].pool()
        self._begin()

    def __enter__(self):
        import gc

        # like torch.cuda.graph: garbage (an earlier runner's graphs, events, tensors) is collected BEFORE the capture - and
        # the collector stays off until it ends: a CUDAGraph or an event destroyed by a collection that happens to trigger
        # while this stream is capturing is a runtime call the capture does not allow
        torch.cuda.synchronize()
        self._gc_was_on = gc.isenabled()
        gc.collect()
        gc.disable()
        torch.cuda.empty_cache()
        self.stream.wait_stream(torch.cuda.current_stream())
        self._ctx = torch.cuda.stream(self.stream)
        self._ctx.__enter__()
        try:
            self._begin()
        except BaseException:
            self._ctx.__exit__(None, None, None)
            if self._gc_was_on:
                gc.enable()
            raise
        return self

    def __exit__(self, et, ev, tb):
        import gc

        try:
            self.graphs[-1].capture_end()
            self.pool = self.graphs[-1].pool()
        finally:
            self._ctx.__exit__(et, ev, tb)
            if self._gc_was_on:
                gc.enable()
        return False


class GraphedRollout:
    """mapper + `policy.act` of one env step as replayable hipGraphs (see the module docstring).  Side effect of the
    split capture: it fixes how the pieces that run beside each other are launched - the gt-semantics mapper narrow
    (`MappingModule.set_launch_width`), the policy's depth encoder as its launch-saving chain when it is the critical
    path and as conv + GroupNorm pairs when RedNet is (`visual_encoder.latency_bound`); eager calls afterwards run the
    same kernels, so replay and eager stay bit-identical."""

    def __init__(self, policy, obs_transforms, example_obs: Dict, deterministic: bool = True, streams: bool = True,
                 warmup: int = 2, extra_keys=(), warmup_mapper: bool = True):
        """warmup_mapper=False: the warm-up steps run everything EXCEPT the mapper's own kernels (MappingModule.dry_run) - for
        a capture in the middle of a rollout whose launch path has changed since the last capture (a persistent encoder was
        retired after a time-out: the launch chain has never run on the capture streams, and its per-stream workspaces must
        exist before a capture can record launches that use them) while the world cloud must not be stepped."""
        self.policy = policy
        self.transforms = list(obs_transforms)
        self._warmup_mapper = warmup_mapper
        self.deterministic = deterministic
        dev = next(policy.parameters()).device
        self.device = dev
        self.static = {k: (v.clone() if torch.is_tensor(v) else v) for k, v in example_obs.items()
                       if (k in _STEP_KEYS or k in extra_keys or not torch.is_tensor(v))}
        # the action kernel writes straight into the next step's previous-action buffer: deterministic steps and
        # sampled steps whose uniforms come with the observations (torch's own Categorical.sample cannot)
        self._direct_actions = deterministic or "_u_sample" in self.static
        B = example_obs["depth"].shape[0]
        H = policy.net._hidden_size
        L = policy.net.num_recurrent_layers
        self.rnn = [torch.zeros(B, L, H, device=dev) for _ in range(2)]
        self.prev = [torch.zeros(B, 1, dtype=torch.long, device=dev) for _ in range(2)]
        self.split = streams == "split"
        self.side = (_stream(dev, "fork_txt"), _stream(dev, "fork_map")) if (streams and not self.split) else None
        self.graphs = []
        self._prefix, self.gA_parts, self._a_tags, self._b_tags = False, [], [], []
        self.phase = 0  # which buffer set holds the current state
        # the fused head's scratch is baked into the captured graphs: this runner owns one (ops.cma_step_ws)
        prev_owner, ops.CMA_WS_OWNER = ops.CMA_WS_OWNER, id(self)
        try:
            if hasattr(policy.net, "prepare_capture"):
                policy.net.prepare_capture(self.static)
            if self.split:
                self._capture_split(warmup)
            else:
                self._capture(warmup)
        finally:
            ops.CMA_WS_OWNER = prev_owner
        # the captured launches hold raw pointers into the instruction encoder's per-shape step caches; the encoder evicts the
        # oldest shape once it has four - this runner keeps the ones that existed at its capture alive
        ienc = getattr(policy.net, "instruction_encoder", None)
        self._held_step_caches = list(getattr(ienc, "__dict__", {}).get("_step_caches", {}).values()) if ienc is not None else []

    def __del__(self):
        try:
            ops.release_cma_ws(id(self))
        except Exception:  # noqa: BLE001 - interpreter shutdown
            pass

    @property
    def actions(self):
        return self.prev[self.phase]

    @property
    def rnn_states(self):
        return self.rnn[self.phase]

    def _body(self, src: int):
        dst = src ^ 1
        cur = torch.cuda.current_stream()
        batch = dict(self.static)
        net = self.policy.net
        if self.side is not None:
            s_txt, s_map = self.side
            s_map.wait_stream(cur)
            # hipGraph replay submits nodes in capture order (~3 us of host time each), so the critical
            # path - the 107-kernel depth ResNet chain - is captured FIRST; the policy then takes the
            # cached `depth_features` path (resnet_encoders.py:92-95)
            with torch.no_grad():
                batch["depth_features"] = net.depth_encoder.visual_encoder(batch)
            with torch.cuda.stream(s_map):  # the mapper feeds only the map CNN, which stays on s_map
                for t in self.transforms:
                    batch = t(batch)
            net._side_streams = self.side
        else:
            for t in self.transforms:
                batch = t(batch)
        net._rnn_out_buffer = self.rnn[dst]
        self.policy._action_out_buffer = self.prev[dst] if self._direct_actions else None
        self._last_batch = batch
        try:
            with torch.no_grad():
                actions, rnn = self.policy.act(batch, self.rnn[src], self.prev[src], _policy_masks(batch),
                                               deterministic=self.deterministic)
                if actions.data_ptr() != self.prev[dst].data_ptr():
                    self.prev[dst].copy_(actions)
                if rnn.data_ptr() != self.rnn[dst].data_ptr():
                    self.rnn[dst].copy_(rnn)
        finally:
            net._side_streams = None
            net._rnn_out_buffer = None
            self.policy._action_out_buffer = None

    # ---- split mode: three graphs per step on two streams ---------------------------------------
    #   gA  (side stream): depth ResNet, its k/v projection, depth_linear   (60 % of the step's kernel
    #                                                      time, depends only on the new depth image)
    #   gB1 (main stream): mapper -> map CNN, instruction encoder, their k/v projections, prev-action embedding
    #   gB2 (main stream, after gA): GRUs, attention, action head
    # Separate graphs on separate streams instead of forked branches inside one capture (which replays
    # slower than a single stream on ROCm 7.2): the only cross-stream edges are two events per step.
    def _capture_split(self, warmup):
        dev = self.device
        net = self.policy.net
        self.ev_in, self.ev_A = torch.cuda.Event(), torch.cuda.Event()
        main = torch.cuda.current_stream()

        B = self.rnn[0].shape[0]
        d_out, m_out = net.depth_linear[1].out_features, net.map_linear[1].out_features
        E, H = net.prev_action_embedding.embedding_dim, net._hidden_size
        self._persist = dict(
            state_in=torch.empty((B, d_out + m_out + E), dtype=torch.float32, device=dev),
            x2=torch.empty((B, H + net.instruction_encoder.output_size + d_out + m_out + E), dtype=torch.float32,
                           device=dev),
        )

        from . import rednet as _rednet

        def run_A():
            net._stage, net._persist = "dep", self._persist
            try:
                with torch.no_grad():
                    batch = dict(self.static)
                    if self._prefix:
                        # the mapper's label-free half (transforms, min / max, keep-highest arg-max: depth and pose only)
                        # rides at the head of the side graph, beside RedNet, instead of behind it on the critical path
                        for t in self.transforms:
                            t.begin_maps(batch)
                        _rednet._stage_done("mapper_begin")
                    net.forward_hip(batch, self.rnn[0], self.prev[0], _policy_masks(batch))
            finally:
                net._stage = net._persist = None

        def run_B1(src):
            batch = dict(self.static)
            for t in self.transforms:
                batch = t.finish_maps(batch) if self._prefix else t(batch)
            net._stage, net._persist = "pre", self._persist
            try:
                with torch.no_grad():
                    net.forward_hip(batch, self.rnn[src], self.prev[src], _policy_masks(batch))
            finally:
                net._stage = net._persist = None
            return batch

        def run_B2(src, batch):
            dst = src ^ 1
            batch = dict(batch)
            net._stage, net._persist = "post", self._persist
            net._rnn_out_buffer = self.rnn[dst]
            self.policy._action_out_buffer = self.prev[dst] if self._direct_actions else None
            try:
                with torch.no_grad():
                    actions, rnn = self.policy.act(batch, self.rnn[src], self.prev[src], _policy_masks(batch),
                                                   deterministic=self.deterministic)
                    if actions.data_ptr() != self.prev[dst].data_ptr():
                        self.prev[dst].copy_(actions)
                    if rnn.data_ptr() != self.rnn[dst].data_ptr():
                        self.rnn[dst].copy_(rnn)
            finally:
                net._stage = net._persist = None
                net._rnn_out_buffer = None
                self.policy._action_out_buffer = None

        # With predicted semantics the depth encoder is NOT the critical path (RedNet is): its launch-saving chain of 16
        # partial slabs per conv would only take HBM bandwidth from RedNet (6.84 vs 6.69 ms per step at 8 envs), so it
        # runs the conv + GroupNorm pairs there.  (Decided before the warm-up: every lazily built cache of the path
        # that will be captured has to exist before the capture.)
        venc = getattr(getattr(net, "depth_encoder", None), "visual_encoder", None)
        predicted = any(getattr(t, "predicted_semantics", False) for t in self.transforms)
        # With predicted semantics gB1 is cut behind one of RedNet's stages (rednet.STAGE_HOOK) and gA is released by an event
        # recorded at the cut: the depth encoder's ~110 small launches then run beside RedNet's pixel-starved deep stages,
        # whose grids leave CUs idle, instead of taking CUs from the chip-filling first ones (round 5: gB1 alone 3242 us,
        # 3562 us with gA started at t = 0; round 6, profiles/r06_split_probe_start.txt: gB1 ends at 3477 / 3461 / 3432 us with
        # the cut behind layer 2 / 3 / 4).  IVLN_PRED_DEPTH_START: stage name, or "0" = no cut (gA starts with the step).
        self._cut_stage = os.environ.get("IVLN_PRED_DEPTH_START", "layer3") if predicted else "0"
        if self._cut_stage in ("0", "", "none"):
            self._cut_stage = None
        # ... and the mapper's label-free half moves to the head of gA (MappingModule.begin / finish, ivln_mapper_step_begin /
        # _finish): gB1 is cut a second time where the labels exist, and waits there for the event recorded behind the prefix
        self._prefix = bool(self._cut_stage) and os.environ.get("IVLN_MAPPER_PREFIX", "1") != "0" and _mapping.STEP_POSED and all(
            hasattr(t, "begin_maps") and type(t).__name__.endswith("IterativeMapper") for t in self.transforms)
        # (the side graph cut in two - its first part beside RedNet's early stages, the second beside its late ones, pausing
        #  during the pixel-starved middle - was built and measured in round 6: gB1 ends at 3453-3462 us against 3408, whatever
        #  the two cut points (profiles/r06_split_probe_twopart.txt): the loss follows the side launches, not the stage they run
        #  beside.  Removed.)
        # the depth encoder's stream: the critical chain of the gt-semantics step wins dispatch when both queues are ready
        # (priority -1; beside RedNet, where it is not critical, the priority made no difference: profiles/r06_split_probe_start.txt)
        self.sA = _stream(dev, "depth", priority=-1)
        if venc is not None:
            venc.latency_bound = not predicted
            venc.beside_other_work = True  # (its graph replays beside the mapper / map-CNN graph: ops.DEPTH_NET's policy)
            # measurement switch for the predicted-semantics step: how the policy's depth encoder runs beside RedNet -
            # "pairs" (default: conv + GroupNorm launches), "chain" (GroupNorm + next conv per launch), "net" (persistent launch)
            pd = os.environ.get("IVLN_PRED_DEPTH", "pairs") if predicted else None
            venc.no_persistent = pd == "chain"
            if pd in ("chain", "net"):
                venc.latency_bound = True
        net._txt_with_dep = predicted  # ... and the instruction encoder leaves RedNet's stream for the side graph
        net._txt_last = True
        # ... and with fewer than 8 images some XCDs stay free of it: the bi-LSTM's blocks that land there take all the work
        # (ops.lstm_bidir spare) instead of half of them waiting for the depth encoder to end
        ienc = net.instruction_encoder
        # (2B * spare blocks so that 2B of them land on the 8 - B free XCDs; measured at 4 and 5 images: 667 -> 620 us per
        #  step; at 6 and 7 - spare 4 and 8 - the recurrence still started only when the depth encoder ended: left alone)
        spare = int(os.environ.get("IVLN_LSTM_SPARE", str(2 if B <= 4 else 3 if B == 5 else 1)))
        if net._txt_last and not predicted and spare > 1 and B < 8:
            # (the captured launches hold this word's ADDRESS: it lives as long as this object's graphs, whatever a later
            #  capture of the same policy hangs on the module)
            self._lstm_ticket = ienc.lstm_ticket = torch.zeros((1,), dtype=torch.int32, device=dev)
            ienc.lstm_spare = spare
        s = _stream(dev, "warmup")
        s.wait_stream(main)
        cap_stream = _stream(dev, "capture")
        with self._dry_mapper():
            with torch.cuda.stream(s):  # warm-up: tables, workspaces (per stream), folded weights
                for i in range(warmup):
                    run_A()
                    run_B2(i & 1, run_B1(i & 1))
            if warmup:
                # ... and once on the streams the graphs are captured on: per-stream state (split-K workspaces, GroupNorm /
                # packed-weight scratch, the depth encoder's arena) must not be born inside a capture - a buffer from a
                # graph's private pool would end up in a process-wide cache
                s.synchronize()
                cap_stream.wait_stream(s)
                with torch.cuda.stream(cap_stream):
                    run_A()
                    run_B2(0, run_B1(0))
                cap_stream.synchronize()
            with torch.cuda.stream(self.sA):  # the side stream needs its own split-K workspace before capture
                run_A()
        main.wait_stream(s)
        torch.cuda.synchronize()
        # A ground-truth-semantics mapper runs beside the depth-ResNet chain with time to spare (gB1 285 us, gA 580):
        # launched narrow it takes ~150 instead of ~60 us but no longer crowds the chain off the CUs (0.767 -> 0.731
        # ms per step at 4 envs).  With predicted semantics RedNet -> mapper IS the critical path: full width.
        # Width: 16 workgroups per env for the local-cloud kernels; the world-cloud kernels may use as many
        # but take 16 points per thread, i.e. 27 workgroups at 109 k points and more only as the cloud grows.
        lw = os.environ.get("IVLN_MAPPER_WIDTH", f"{16 * B},{16 * B}")
        for t in self.transforms:
            mm = getattr(t, "mapping_module", None)
            if mm is not None and hasattr(mm, "set_launch_width"):
                narrow = getattr(mm, "semantics_module", None) is None and lw != "0"
                mm.set_launch_width(*([int(v) for v in lw.split(",")] if narrow else [0, 0]))
        ops.settle_packed_weights()
        self.ev_pre = torch.cuda.Event()
        with _CutCapture(self.sA) as ca:
            a_tags = []

            def hook_a(name, ca=ca):
                if name == "mapper_begin" and not a_tags:
                    a_tags.append("pre")
                    ca.cut()
            prev_hook, _rednet.STAGE_HOOK = _rednet.STAGE_HOOK, (hook_a if self._prefix else None)
            try:
                run_A()
            finally:
                _rednet.STAGE_HOOK = prev_hook
        # [mapper prefix | depth encoder (+ instruction encoder)], or the one graph: a_tags names the boundary behind each piece but the last
        self.gA_parts, self._a_tags = list(ca.graphs), a_tags
        self.gA = self.gA_parts[-1]
        self._dep = net._stash_dep  # (depth features, k/v) live in gA's pool
        self._txt = getattr(net, "_stash_txt", None)  # (predicted semantics: the instruction features too)
        # the previous action (read by the embedding in gB1) ping-pongs with the state: one gB1 per phase
        self.ev_mid = torch.cuda.Event()
        self.gB1, pool = [], None
        for src in (0, 1):
            with _CutCapture(cap_stream, pool) as cc:
                b_tags = []

                def hook(name, cc=cc, b_tags=b_tags):
                    if name == self._cut_stage and "mid" not in b_tags:
                        b_tags.append("mid")
                        cc.cut()
                    elif name == "labels" and self._prefix and "labels" not in b_tags and "mid" in b_tags:
                        b_tags.append("labels")
                        cc.cut()  # (the mapper's second half starts here: the replay waits for the prefix's event in between)
                prev_hook, _rednet.STAGE_HOOK = _rednet.STAGE_HOOK, (hook if self._cut_stage else None)
                try:
                    net._stash_txt = self._txt
                    batch = run_B1(src)
                finally:
                    _rednet.STAGE_HOOK = prev_hook
            pool = cc.pool
            stash = net._stash
            g2 = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g2, pool=pool, stream=cap_stream):
                net._stash, net._stash_dep = stash, self._dep
                run_B2(src, batch)
            self.gB1.append(list(cc.graphs))
            self._b_tags = b_tags  # (the same cuts in both phases)
            self.graphs.append(g2)
        self._keep = (batch, stash)
        self.phase = 0
        if venc is not None:
            venc.beside_other_work = False  # (the captured launches are what they are; eager calls decide for themselves)
        ienc.lstm_spare = 1  # (the ticket word stays with the module: the captured launches hold its address)

    def _replay_split(self, mark=None):
        """One step: the main graphs on the current stream, the side graph(s) on sA.  `mark(name, stream)` (tools/split_probe.py)
        is called where gA starts / ends and where gB1 ends, to record timing events.
        (Tried and left: gB1 BEFORE gA - no difference, round 3; gB1 on a third, CU-masked stream - 0.79-0.83 vs 0.708 ms per
        gt step, round 3; gB2 behind gA on the side stream - slower; the main graphs on a third stream with a priority of its
        own - 8.2 vs 4.25 ms, round 5: the extra stream hop serialises the replay.)"""
        main = torch.cuda.current_stream()
        pieces = self.gB1[self.phase]
        b_tags = getattr(self, "_b_tags", []) if len(pieces) > 1 else []
        a_parts, a_tags = self.gA_parts, self._a_tags

        def side(wait_ev):
            """the side graph's pieces behind `wait_ev`; the event behind the mapper's prefix is recorded where it ends"""
            self.sA.wait_event(wait_ev)
            with torch.cuda.stream(self.sA):
                if mark:
                    mark("gA_start", self.sA)
                for i, g in enumerate(a_parts):
                    g.replay()
                    if i < len(a_tags) and a_tags[i] == "pre":
                        self.ev_pre.record(self.sA)
                if mark:
                    mark("gA_end", self.sA)
                self.ev_A.record(self.sA)

        if not b_tags:  # one main graph: the side graph starts with the step
            self.ev_in.record(main)
            side(self.ev_in)
            if "pre" in a_tags:  # (a mapper prefix in the side graph but no cut in the main one: it has to be over first)
                main.wait_event(self.ev_pre)
            pieces[-1].replay()
        else:
            for i, g in enumerate(pieces):
                tag = b_tags[i - 1] if i > 0 else None  # the boundary in FRONT of this piece
                if tag == "mid":      # ... released here: the side graph
                    self.ev_mid.record(main)
                    side(self.ev_mid)
                elif tag == "labels":  # the mapper's second half waits for its first (the side graph's head)
                    main.wait_event(self.ev_pre)
                g.replay()
        if mark:
            mark("gB1_end", main)
        main.wait_event(self.ev_A)
        self.graphs[self.phase].replay()

    def _dry_mapper(self):
        """Context: the transforms' mappers skip their own kernels (warmup_mapper=False), nothing otherwise."""
        import contextlib

        mms = [] if self._warmup_mapper else [mm for mm in (getattr(t, "mapping_module", None) for t in self.transforms) if mm is not None]

        @contextlib.contextmanager
        def ctx():
            for mm in mms:
                mm.dry_run = True
            try:
                yield
            finally:
                for mm in mms:
                    mm.dry_run = False

        return ctx()

    def _capture(self, warmup):
        s = _stream(self.device, "warmup")
        s.wait_stream(torch.cuda.current_stream())
        with self._dry_mapper(), torch.cuda.stream(s):  # warm-up off the default stream: creates tables, workspaces, handles
            for i in range(warmup):
                self._body(i & 1)
        torch.cuda.current_stream().wait_stream(s)
        torch.cuda.synchronize()
        ops.settle_packed_weights()
        # captured on the stream the warm-up ran on: per-stream resources (split-K workspaces, the persistent depth
        # encoder's arena and sync words) exist for it - a capture stream of torch's own would find none and could not
        # create them inside the capture
        for src in (0, 1):
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, stream=s):
                self._body(src)
            self.graphs.append(g)
        self.phase = 0

    def load(self, obs: Dict):
        """Observation tensors -> the captured input buffers: one multi-copy launch for everything that is
        already a matching contiguous device tensor, Tensor.copy_ (H2D / cast / strided) for the rest."""
        pairs = []
        for k, v in obs.items():
            if torch.is_tensor(v):
                dst = self.static.get(k)
                if dst is None:
                    continue
                if v.is_cuda and v.dtype == dst.dtype and v.shape == dst.shape and v.is_contiguous():
                    pairs.append((v, dst))
                else:
                    dst.copy_(v, non_blocking=True)
            else:
                self.static[k] = v
        if pairs:
            ops.copy_multi(pairs)

    def step(self, obs: Dict = None):
        """Copy the observations the step reads into the captured input buffers and replay.
        Returns the (B,1) int64 action tensor (one of the two persistent buffers)."""
        if obs is not None:
            self.load(obs)
        if self.split:
            self._replay_split()
        else:
            self.graphs[self.phase].replay()
        self.phase ^= 1
        return self.actions

    def redo_last_step_eagerly(self):
        """The policy half of the LAST replayed step again, as eager launches, from the same inputs and the same recurrent
        state: what a loop does after a persistent kernel of the step timed out (depth_net.any_failed -> recover_all, which
        retires the persistent form: these launches take the launch chain).  The mapper's part of the step stands - it does
        not depend on the depth encoder and its maps sit in the mapper's persistent buffers.  BatchNorm running statistics
        (a policy in train mode: quirk Q6) were already updated by the replayed step and are not updated again.  Returns the
        action tensor; `rnn_states` / `actions` then hold the recomputed values.  The runner must not be replayed afterwards
        (its graphs still contain the retired launch): callers capture a new one."""
        src, dst = self.phase ^ 1, self.phase  # (step() has flipped the phase)
        batch = dict(self.static)
        occ, sem = self.maps()
        if occ is not None:
            batch["occupancy_map"], batch["semantic_map"] = occ, sem
        bns = [m for m in self.policy.modules() if isinstance(m, torch.nn.modules.batchnorm._BatchNorm)]
        saved = [(m.momentum, m.num_batches_tracked.clone() if m.num_batches_tracked is not None else None) for m in bns]
        for m in bns:
            m.momentum = 0.0
        try:
            with torch.no_grad():
                actions, rnn = self.policy.act(batch, self.rnn[src], self.prev[src], _policy_masks(batch),
                                               deterministic=self.deterministic)
                self.prev[dst].copy_(actions)
                self.rnn[dst].copy_(rnn)
        finally:
            for m, (mom, nbt) in zip(bns, saved):
                m.momentum = mom
                if nbt is not None:
                    m.num_batches_tracked.copy_(nbt)
        return self.actions

    def maps(self):
        """(occupancy_map, semantic_map) of the last replayed step: the mapper's persistent (B, 64, 64) u8 buffers, as
        the captured batch holds them (None, None without a mapper)."""
        batch = self._keep[0] if self.split else getattr(self, "_last_batch", {})
        return batch.get("occupancy_map"), batch.get("semantic_map")

    def depth_features(self):
        """(B, 128, 4, 4) output of `net.depth_encoder.visual_encoder` of the last replayed step - what the collection
        loops cache per step (dagger_trainer.py:317-323; a forward hook cannot fire inside a replay): the leading
        channels of the depth branch's (B, 192, 4, 4) buffer in the side graph's pool."""
        if not self.split:
            return None
        dep = self._dep[0]
        return dep[:, : self.policy.net.depth_encoder.visual_encoder.output_shape[0]]

    def reset_state(self):
        for t in self.rnn + self.prev:
            t.zero_()
        self.phase = 0
