"""PointNet++(MSG) encoder on the HIP kernels - drop-in for `Pointnet2ClsMSG(0)` (networks/pts_encoder/pointnet2.py:166-211).

forward(pts [B,N,3] f32 on the GPU) -> [B,1024].  Launch sequence per call (all on torch's current stream):
  1 x gp_fps_chain      FPS + gather for every level (one workgroup per cloud)
  L x gp_ball_query_msg both radii of a level in one pass
  L x gp_point_linear   hoisted feature half of the first layer, once per source point (both scales)
  2L x gp_sa_pre_mlp_max_layout  gather -> xyz half of layer 1 -> layers 2-3 on fp32 MFMA -> max-pool, per scale (register-chain kernels;
                        the hidden-layer layout the weights were packed for, weights.SAScale)
  1 + 2 x the same pair  GroupAll level: whole rounds of 256 clouds on the ring kernel (one cloud per workgroup), the rest on 32-row tiles
                        (the tiles of a cloud combine by integer atomic max into a zeroed buffer)
Intermediate features stay point-major [B, n, C]; the reference's grouped [B,C+3,np,ns] tensors never exist.
"""
import ctypes
import itertools
import weakref

import torch

from . import _lib
from ._lib import ptr, stream_ptr
from .config import dist_arith_code
from .lru import ShapeCache
from .weights import EncoderWeights


_GENERATION = itertools.count(1)  # process-wide: every write of a workspace's grouping buffers gets a generation no other write has


def _unpinned(ws):
    return ws.get("_pins", 0) == 0


def _unpin_entry(key, ent):
    """A captured pass leaves the cache: the workspaces its replays wrote into return to the eviction order."""
    for ws in ent.get("pinned", ()):
        ws["_pins"] = ws.get("_pins", 1) - 1


class Pointnet2EncoderHIP:
    def __init__(self, state_dict, device="cuda", params="light", prefix="pts_encoder.", arith=None, precision="f32"):
        """arith: contraction convention of the squared distances in furthest point sampling and the ball queries ('A' | 'B' | 'C',
        config.DEFAULT_DIST_ARITH when None; include/genpose_hip.h GP_ARITH_*).
        precision: 'f32' (default: every dense layer on the fp32 matrix pipe - what all parity claims and the headline bench line run) or
        'bf16x3' (OPT-IN, exploratory, round 5): grouping levels 1 and 2 (64-64/96-128, 128-196-256) on the bf16 matrix pipe as three-term split products with
        fp32 accumulation (csrc/sa_bf16x3.hip; ~2^-17 relative per product); centres and neighbourhoods are unaffected."""
        if precision not in ("f32", "bf16x3"):
            raise ValueError(f"encoder precision {precision!r}: 'f32' or 'bf16x3'")
        self.precision = precision
        self.device = torch.device(device)
        self.arith = dist_arith_code(arith)
        self.w = EncoderWeights(state_dict, self.device, params, prefix)
        self.cfg = self.w.cfg
        self.out_dim = self.w.out_dim
        # workspaces (~1.5 MB per cloud) per (batch, points, slot): least recently used first, except those a captured graph writes
        # into (`_pins`: taken by whoever captures, given back when that graph is dropped)
        self._ws = ShapeCache(self.MAX_WORKSPACES, can_evict=_unpinned)
        self._pass_graphs = ShapeCache(self.MAX_PASS_GRAPHS, on_evict=_unpin_entry)  # (no closure over self: no reference cycle)
        self._seen_once = ShapeCache(4 * self.MAX_PASS_GRAPHS)

    MAX_WORKSPACES = 12
    # (layer widths, neighbourhood size) the opt-in split-bf16 kernel is instantiated for: grouping levels 1 and 2 (csrc/sa_bf16x3.hip)
    BF16X3_SHAPES = {((128, 196, 256), 16), ((128, 196, 256), 32), ((64, 64, 128), 16), ((64, 96, 128), 32), ((64, 64, 128), 32)}

    def pin_workspaces(self, B, N, slot=0):
        """For whoever CAPTURES launches of this encoder in a hipGraph: the workspace of that geometry stays out of the eviction order
        while the graph lives (its buffers were allocated outside the capture; nothing but this cache holds them).  Returns the
        workspace; give the pin back with `ws['_pins'] -= 1` when the graph is dropped."""
        ws = self._workspace(B, N, slot)
        ws["_pins"] = ws.get("_pins", 0) + 1
        return ws

    def _wait_pending_join(self, ws):
        """A deferred grouping (prepare_grouping(defer_join=True)) whose consumer never ran - an exception in between, a caller that only
        wanted the ticket - leaves work on the side stream that reads new_xyz[0] and writes the deeper levels: every writer of the
        grouping buffers waits for it first."""
        pending = ws.pop("_join", None)
        if pending:
            for ev in pending.values():  # {level: event}
                torch.cuda.current_stream(self.device).wait_event(ev)

    # ------------------------------------------------------------------ workspace (cached per batch/size)
    def _workspace(self, B, N, slot=0):
        key = (B, N, slot)
        ws = self._ws.get(key)
        if ws is not None:
            return ws
        dev = self.device
        ws = {"fps_idx": [], "new_xyz": [], "bq": [], "feat": [], "z": []}
        n = N
        cin = 0
        for k, npnt in enumerate(self.cfg["npoints"]):
            cout = sum(s.couts[-1] for s in self.w.levels[k])
            if npnt is None:
                ws["feat"].append(torch.zeros(B, 1, cout, device=dev))
                ws["z"].append(torch.empty(B, n, sum(s.couts[0] for s in self.w.levels[k]), device=dev) if cin > 0 else None)
                break
            ws["fps_idx"].append(torch.empty(B, npnt, dtype=torch.int32, device=dev))
            ws["new_xyz"].append(torch.empty(B, npnt, 3, device=dev))
            ws["bq"].append([torch.empty(B, npnt, ns, dtype=torch.int32, device=dev) for ns in self.cfg["nsamples"][k]])
            ws["feat"].append(torch.empty(B, npnt, cout, device=dev))
            zs = sum(s.couts[0] for s in self.w.levels[k])
            ws["z"].append(torch.empty(B, n, zs, device=dev) if cin > 0 else None)
            n, cin = npnt, cout
        self._ws[key] = ws
        return ws

    def sample_centres(self, pts, slot=0):
        """Stage 1 of forward(): furthest point sampling + gather for every level into workspace `slot` (launches on the
        current stream).  Latency-bound and light (one workgroup per cloud), so a pipeline can run it for the NEXT batch on a
        side stream while the MFMA-heavy stages of the current one own the chip; forward(..., slot=slot, centres_done=True)
        then skips it."""
        _lib.check_device()
        if not pts.is_cuda or pts.dtype != torch.float32:
            raise RuntimeError("pts must be a float32 CUDA tensor")
        xyz0 = pts[..., 0:3].contiguous()
        B, N, _ = xyz0.shape
        ws = self._workspace(B, N, slot)
        self._wait_pending_join(ws)
        ws["_gen"] = next(_GENERATION)  # fps_idx / new_xyz are about to be overwritten: tickets for the previous contents die here
        st = stream_ptr()
        cfg = self.cfg
        group_levels = [k for k, npnt in enumerate(cfg["npoints"]) if npnt is not None]
        if len(group_levels) <= 3 and N <= 1024:
            m = (ctypes.c_int * 3)(*([cfg["npoints"][k] for k in group_levels] + [0] * (3 - len(group_levels))))
            pi = [ptr(ws["fps_idx"][l]) if l < len(group_levels) else None for l in range(3)]
            px = [ptr(ws["new_xyz"][l]) if l < len(group_levels) else None for l in range(3)]
            _lib.call("gp_fps_chain_arith", self.arith, B, N, len(group_levels), m, ptr(xyz0), pi[0], px[0], pi[1], px[1], pi[2], px[2], st)
        else:
            cur = xyz0
            for l, k in enumerate(group_levels):
                npnt = cfg["npoints"][k]
                temp = torch.full((B, cur.shape[1]), 1e10, device=self.device)
                _lib.call("gp_furthest_point_sampling_arith", self.arith, B, cur.shape[1], npnt, ptr(cur), ptr(temp), ptr(ws["fps_idx"][l]), st)
                torch.gather(cur, 1, ws["fps_idx"][l].long().unsqueeze(-1).expand(B, npnt, 3), out=ws["new_xyz"][l])
                cur = ws["new_xyz"][l]
        return xyz0

    def _ball_queries(self, ws, xyz0, B, N, levels=None):
        """Ball queries of every grouping level (they depend on the coordinates only) into ws['bq']; `levels`: only those levels."""
        if levels is None:  # (a level-restricted call is prepare_grouping's own, which has just handled the pending join)
            self._wait_pending_join(ws)
        ws["_gen"] = next(_GENERATION)  # every writer of the grouping buffers invalidates outstanding tickets
        st = stream_ptr()
        cfg = self.cfg
        xyz, n = xyz0, N
        for k, npnt in enumerate(cfg["npoints"]):
            if npnt is None:
                break
            new_xyz = ws["new_xyz"][k]
            if levels is not None and k not in levels:
                xyz, n = new_xyz, npnt
                continue
            radii, nss = cfg["radii"][k], cfg["nsamples"][k]
            if len(self.w.levels[k]) == 2:
                _lib.call("gp_ball_query_msg_arith", self.arith, B, n, npnt, float(radii[0]), nss[0], float(radii[1]), nss[1], ptr(new_xyz), ptr(xyz),
                          ptr(ws["bq"][k][0]), ptr(ws["bq"][k][1]), st)
            else:
                for i in range(len(self.w.levels[k])):
                    ws["bq"][k][i].zero_()
                    _lib.call("gp_ball_query_arith", self.arith, B, n, npnt, float(radii[i]), nss[i], ptr(new_xyz), ptr(xyz), ptr(ws["bq"][k][i]), st)
            xyz, n = new_xyz, npnt

    def grouping_key(self):
        """Everything the sampled centres and neighbourhoods depend on besides the coordinates."""
        c = self.cfg
        return (tuple(c["npoints"]), tuple(map(tuple, c["radii"])), tuple(map(tuple, c["nsamples"])), self.arith)

    def prepare_grouping(self, pts, slot=0, defer_join=False):
        """Furthest point sampling + gather + ball queries for every level into workspace `slot`; returns that workspace.  These
        depend on the point coordinates only - not on any weight - so a second encoder with the same grouping configuration (the
        energy model's, which sees the same clouds) can take them over: forward(pts, grouping=<this workspace>).

        defer_join (the encoder pass of an agent call): furthest point sampling is a chain of dependent block-wide argmax steps - 57 % of
        them belong to level 0 - that occupies one wave per SIMD and no matrix pipe.  Only level 0 is sampled on the calling stream;
        the deeper levels and their ball queries go to a side stream and run UNDERNEATH the level-0 set abstraction (ball query, both
        chain kernels, the hoisted GEMM of level 1).  The workspace then carries the join event (`_join`), which forward(grouping=ws)
        waits for before its first use of a deeper level."""
        self._wait_pending_join(self._workspace(pts.shape[0], pts.shape[1], slot))  # a deferred grouping nobody consumed
        cfg = self.cfg
        group_levels = [k for k, npnt in enumerate(cfg["npoints"]) if npnt is not None]
        if not (defer_join and 2 <= len(group_levels) <= 3 and pts.shape[1] <= 1024 and pts.is_cuda):
            xyz0 = self.sample_centres(pts, slot)
            B, N, _ = xyz0.shape
            ws = self._workspace(B, N, slot)
            self._ball_queries(ws, xyz0, B, N)
            ws["_grouping_key"] = self.grouping_key()
            return ws
        _lib.check_device()
        if pts.dtype != torch.float32:
            raise RuntimeError("pts must be a float32 CUDA tensor")
        xyz0 = pts[..., 0:3].contiguous()
        B, N, _ = xyz0.shape
        ws = self._workspace(B, N, slot)
        ws["_gen"] = next(_GENERATION)
        cur = torch.cuda.current_stream(self.device)
        if getattr(self, "_side", None) is None:
            self._side = torch.cuda.Stream(self.device)
        m0 = (ctypes.c_int * 3)(cfg["npoints"][group_levels[0]], 0, 0)
        _lib.call("gp_fps_chain_arith", self.arith, B, N, 1, m0, ptr(xyz0), ptr(ws["fps_idx"][0]), ptr(ws["new_xyz"][0]), None, None, None, None, stream_ptr())
        fork = torch.cuda.Event()
        fork.record(cur)
        joins = {}
        with torch.cuda.stream(self._side):
            self._side.wait_event(fork)
            # the deeper levels select among the previous level's centres, in their order (the reference samples new_xyz of the level
            # before): ONE level per launch, each followed by its ball query and its own join event, so that the set abstraction of level k
            # starts as soon as ITS centres and neighbourhoods exist - level 1's kernels run beside level 2's sampling (round 6: at 5 clouds
            # the fused two-level chain held the pass for 97 us of sampling + both ball queries; a level at a time it is 60 us + one)
            n_src = cfg["npoints"][group_levels[0]]
            for l, k in enumerate(group_levels[1:]):
                mk = (ctypes.c_int * 3)(cfg["npoints"][k], 0, 0)
                _lib.call("gp_fps_chain_arith", self.arith, B, n_src, 1, mk, ptr(ws["new_xyz"][l]), ptr(ws["fps_idx"][l + 1]), ptr(ws["new_xyz"][l + 1]), None, None,
                          None, None, stream_ptr())
                self._ball_queries(ws, xyz0, B, N, levels={k})
                joins[k] = torch.cuda.Event()
                joins[k].record(self._side)
                n_src = cfg["npoints"][k]
        self._ball_queries(ws, xyz0, B, N, levels={group_levels[0]})
        ws["_join"] = joins  # {level: event}
        ws["_grouping_key"] = self.grouping_key()
        return ws

    def grouping_ticket(self, pts, ws):
        """What a second encoder needs to take this grouping over safely (GFObjectPose.extract_pts_feature): WHICH tensor object the
        clouds are (a weak reference - an address can be reused by the allocator, an object cannot), which version of it (in-place
        edits bump `_version`), and which generation of the workspace (every writer of the grouping buffers - prepare_grouping, a plain
        forward(), sample_centres - takes a fresh process-wide generation, so a ticket never outlives the contents it was issued for)."""
        return {"ws": ws, "gen": ws["_gen"], "key": self.grouping_key(), "pts": weakref.ref(pts), "version": pts._version, "shape": tuple(pts.shape)}

    @staticmethod
    def ticket_valid(ticket, pts, key):
        return (ticket is not None and ticket["key"] == key and ticket["pts"]() is pts and ticket["version"] == pts._version
                and ticket["shape"] == tuple(pts.shape) and ticket["ws"].get("_gen") == ticket["gen"])

    MAX_PASS_GRAPHS = 8  # captured encoder passes kept per encoder (least recently used first)

    def encode(self, pts, grouping=None, use_graph=True):
        """The whole encoder pass of an AGENT call (GFObjectPose.extract_pts_feature) as one hipGraph replay per input shape:
            grouping None  ->  prepare_grouping + forward: returns (feat [B,1024], workspace) - the workspace is what a second encoder takes
                               over (grouping_ticket);
            grouping = ws  ->  forward on those centres and neighbourhoods: returns feat.
        ~25 launches replayed instead of issued (they dominate a small batch: a 5-cloud pass is launch-bound).  The first call of a shape
        runs launch by launch (warm-up; a shape seen once never pays a capture), the second captures, later ones copy the clouds into the
        graph's static input and replay.  Results are the launch-by-launch pass's, bit for bit (same kernels, same order)."""
        direct = (lambda: self.forward(pts, grouping=grouping)) if grouping is not None else (lambda: self._forward_with_grouping(pts))
        if not use_graph or not pts.is_cuda or pts.dtype != torch.float32 or torch.cuda.is_current_stream_capturing():
            return direct()  # (wrong device / dtype: the launch-by-launch path raises what it always raised)
        # (the id() of `grouping` is part of the key: an entry with a graph keeps its `grouping` alive, so the id cannot be reused
        # while the entry lives; a shape only SEEN so far is remembered by shape alone)
        key = (tuple(pts.shape), id(grouping) if grouping is not None else None)
        ent = self._pass_graphs.get(key)
        if ent is None:
            seen = (tuple(pts.shape), grouping is not None)
            if seen not in self._seen_once:
                # first call of a shape: launch by launch (a shape that never returns never pays a capture, and - kept apart from the
                # captured graphs - never evicts one: with more shapes in rotation than the cache holds, a shape is simply captured
                # again when it comes back, instead of every newcomer throwing out a live graph)
                self._seen_once[seen] = True
                return direct()
            buf = pts[..., 0:3].contiguous().clone()
            B, N = buf.shape[0], buf.shape[1]
            # the workspace the replay writes into is pinned BEFORE the capture: were it evicted between the first (direct) call and this
            # one, _workspace() would otherwise allocate it inside torch.cuda.graph - in the graph's private pool - while the cache holds it
            pinned = [self.pin_workspaces(B, N, 0)]
            torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            try:
                with torch.cuda.graph(g):
                    out = self.forward(buf, grouping=grouping) if grouping is not None else self._forward_with_grouping(buf)
            except BaseException:
                for ws in pinned:
                    ws["_pins"] -= 1
                raise
            # `grouping` is kept alive (its id() is part of the key)
            ent = {"graph": g, "buf": buf, "out": out, "grouping": grouping, "pinned": pinned}
            self._pass_graphs[key] = ent
        ent["buf"].copy_(pts[..., 0:3])
        if grouping is None:
            ent["out"][1]["_gen"] = next(_GENERATION)  # the replay rewrites the grouping buffers: tickets for the previous contents die here
            ent["graph"].replay()
            return ent["out"][0].clone(), ent["out"][1]
        ent["graph"].replay()
        return ent["out"].clone()

    def _forward_with_grouping(self, pts):
        ws = self.prepare_grouping(pts, defer_join=True)
        return self.forward(pts, grouping=ws), ws

    def forward(self, pts, return_intermediates=False, slot=0, centres_done=False, grouping=None):
        """grouping: workspace returned by prepare_grouping() of an encoder with the same grouping configuration, for the SAME
        clouds: its centres and neighbourhood indices are used instead of being recomputed."""
        _lib.check_device()
        if not pts.is_cuda or pts.dtype != torch.float32:
            raise RuntimeError("pts must be a float32 CUDA tensor")
        xyz0 = pts[..., 0:3].contiguous()
        B, N, _ = xyz0.shape
        ws = self._workspace(B, N, slot)
        st = stream_ptr()
        cfg = self.cfg
        if grouping is not None:
            if grouping.get("_grouping_key") != self.grouping_key() or grouping["new_xyz"][0].shape[0] != B:
                raise ValueError("grouping comes from an encoder with a different configuration / batch")
            src = grouping
        else:
            src = ws
            if not centres_done:
                self._wait_pending_join(ws)
            # ---- furthest point sampling + gather for every level
            if not centres_done:
                self.sample_centres(pts, slot)
            self._ball_queries(ws, xyz0, B, N)
        # ---- set abstraction levels
        xyz, feats, n, cin = xyz0, None, N, 0
        for k, npnt in enumerate(cfg["npoints"]):
            scales = self.w.levels[k]
            out = ws["feat"][k]
            cout_total = out.shape[-1]
            if npnt is None:
                # GroupAll: same hoisted form - layer 1's feature half is one GEMM over all B*n points (both scales), the
                # tiles of a cloud combine through an integer atomic max into the zeroed output
                out.zero_()
                z = ws["z"][k]
                zstride = sum(sc.couts[0] for sc in scales)
                if z is not None:
                    _lib.call("gp_point_linear", B * n, cin, zstride, ptr(feats), ptr(self.w.z_weights[k]), ptr(z), st)
                off, zoff = 0, 0
                for sc in scales:
                    (w1, b1), (w2, b2), (w3, b3) = sc.layers
                    _lib.call("gp_sa_pre_mlp_max_layout", sc.hidden_layout, B, n, 1, n, sc.couts[0], sc.couts[1], sc.couts[2], ptr(xyz), None, None, ptr(z), zstride,
                              zoff, ptr(sc.wxyz), ptr(b1), ptr(w2), ptr(b2), ptr(w3), ptr(b3), ptr(out), cout_total, off, st)
                    off += sc.couts[2]
                    zoff += sc.couts[0]
                feats, n, cin = out, 1, cout_total
                break
            new_xyz = src["new_xyz"][k]
            nss = cfg["nsamples"][k]
            # first layer hoisted: feature half once per source point, xyz half while gathering (csrc/sa_mlp.hip)
            z = ws["z"][k]
            zstride = sum(sc.couts[0] for sc in scales)
            if z is not None:
                _lib.call("gp_point_linear", B * n, cin, zstride, ptr(feats), ptr(self.w.z_weights[k]), ptr(z), st)
            if k >= 1 and src.get("_join") and k in src["_join"]:
                # deferred grouping (prepare_grouping(defer_join=True)): this level's centres and neighbourhoods were computed on the
                # side stream under the levels before it - first use here
                torch.cuda.current_stream(self.device).wait_event(src["_join"].pop(k))
                if not src["_join"]:
                    src.pop("_join")
            off, zoff = 0, 0
            for i, sc in enumerate(scales):
                (w1, b1), (w2, b2), (w3, b3) = sc.layers
                if (self.precision == "bf16x3" and z is not None and (tuple(sc.couts), nss[i]) in self.BF16X3_SHAPES
                        and (B * npnt * nss[i]) % 32 == 0):
                    w2s, b2s, w3s, b3s = sc.bf16x3_packs()
                    _lib.call("gp_sa_pre_mlp_max_bf16x3", B, n, npnt, nss[i], sc.couts[0], sc.couts[1], sc.couts[2], ptr(xyz), ptr(new_xyz), ptr(src["bq"][k][i]), ptr(z),
                              zstride, zoff, ptr(sc.wxyz), ptr(b1), ptr(w2s), ptr(b2s), ptr(w3s), ptr(b3s), ptr(out), cout_total, off, st)
                    off += sc.couts[2]
                    zoff += sc.couts[0]
                    continue
                _lib.call("gp_sa_pre_mlp_max_layout", sc.hidden_layout, B, n, npnt, nss[i], sc.couts[0], sc.couts[1], sc.couts[2], ptr(xyz), ptr(new_xyz),
                          ptr(src["bq"][k][i]), ptr(z), zstride, zoff, ptr(sc.wxyz), ptr(b1), ptr(w2), ptr(b2), ptr(w3), ptr(b3), ptr(out),
                          cout_total, off, st)
                off += sc.couts[2]
                zoff += sc.couts[0]
            xyz, feats, n, cin = new_xyz, out, npnt, cout_total
        res = feats.reshape(B, -1).clone()
        if return_intermediates:
            if src is not ws:
                ws = dict(ws, new_xyz=src["new_xyz"], fps_idx=src["fps_idx"], bq=src["bq"])
            return res, ws
        return res

    __call__ = forward
