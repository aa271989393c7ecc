"""Temporal history fusion of FB-OCC as a standalone module.

Mirror of the history part of `FBOCC` -- mmdet3d/models/fbbev/detectors/fbocc.py:
layers and state :101-131, `generate_grid` :169-205, `fuse_history` :207-319 -- with the same constructor
argument names, the same sub-module names (`history_keyframe_time_conv`, `history_keyframe_cat_conv`: a detector
state_dict loads unchanged) and the same state semantics (`history_bev`, `history_seq_ids`,
`history_forward_augs`, `history_sweep_time`, sequence restarts, `do_history`).

What runs where:
  * rt_flow and the trilinear warp of the T-frame history: HIP (`fbbev_history_flow`, `fbbev_history_warp`) -- the
    sampling grid is never materialised;
  * inference: the warp writes straight into the frame slots 1..T of the next (T+1)-frame buffer, the current frame
    is copied into slot 0, and `history_bev` becomes a VIEW of slots 0..T-1 (the reference cats, clones and re-cats
    the 16x80-channel volume: ~6 full passes); the time channel of the 81->80 conv is folded into a per-(sample,
    frame) bias and the eval-mode batch norms into the 1x1x1 conv weights; both convs then run as ONE fp32-MFMA kernel
    (`fbbev_history_conv`: per 64-voxel tile, relu(W1 x_t + b_t) is parked in LDS and immediately consumed by the
    W2_t accumulation -- the 1360-channel intermediate never reaches HBM); channel counts that are not multiples of
    16 fall back to two batched library GEMMs with a bias epilogue;
  * training (grad enabled): the reference's own op sequence on the module's layers (batch-norm statistics intact),
    only the warp is the HIP kernel -- `history_bev` is detached (:241), so the warp needs no backward.
No CPU fallback: the HIP extension must be present and the tensors on the GPU.
"""
import torch
import torch.nn as nn
import torch.nn.functional as F

from . import _capi
from .mfma_conv3d import MConv3d      # nn.Conv3d (same parameters / state dict) that can take the fbbev_conv3d_* autograd route


class TemporalHistoryFusion(nn.Module):
    def __init__(self, dx, bx, single_bev_num_channels=80, history_cat_num=16, history_cat_conv_out_channels=None,
                 do_history=True, interpolation_mode='bilinear', history_cam_sweep_freq=0.5, history_dtype=torch.float32,
                 history_compute='auto', ring_layout='voxel_major'):
        super().__init__()
        if interpolation_mode != 'bilinear':
            raise NotImplementedError("only interpolation_mode='bilinear' (trilinear on the voxel grid) is built")
        self.dx = [float(v) for v in dx]                               # forward_projection.dx  (fbocc.py:186-188)
        self.lower = [float(b) - float(d) / 2.0 for b, d in zip(bx, dx)]   # bx - dx/2           (fbocc.py:189-191)
        self.single_bev_num_channels = C = single_bev_num_channels
        self.do_history = do_history
        self.interpolation_mode = interpolation_mode
        self.history_cat_num = T = history_cat_num
        self.history_cam_sweep_freq = history_cam_sweep_freq           # seconds between frames (fbocc.py:105)
        out_c = history_cat_conv_out_channels if history_cat_conv_out_channels is not None else C
        self.history_keyframe_time_conv = nn.Sequential(               # fbocc.py:111-118
            MConv3d(C + 1, C, kernel_size=1, padding=0, stride=1), nn.SyncBatchNorm(C), nn.ReLU(inplace=True))
        self.history_keyframe_cat_conv = nn.Sequential(                # fbocc.py:120-127
            MConv3d(C * (T + 1), out_c, kernel_size=1, padding=0, stride=1), nn.SyncBatchNorm(out_c),
            nn.ReLU(inplace=True))
        self.use_mfma_convs = True          # inference: both 1x1x1 convs in one fp32-MFMA kernel when the channels allow
        self.fused_x3 = False               # 16-bit voxel-major ring, split-operand convolutions, C = 80: the whole step in one kernel
                                            # (fbbev_history_fused_x3_vm)
        self.pipelined_step = False         # opt-in: the warp of a band of rows and the split-operand convolutions of the band before it on
        self.pipelined_step_chunks = 0      # two streams (fbbev_history_step_x3_vm).  Same bits; measured NOT faster (the convolutions' workgroups
                                            # starve while warp workgroups wait for CUs: profiles/r06_exp_history_step.md)
        self.fused_warp_conv = False        # opt-in.  inference, 16-bit voxel-major ring + bf16 convolutions (C = 80): warp and
                                            # both convolutions in ONE kernel (fbbev_history_fused_vm).  Bit-identical to the
                                            # two-kernel path and 2.1 GB less HBM read per step at 400x400x16, but measured
                                            # SLOWER (6.8 vs 5.6 ms: DESIGN 7, profiles/r03_hist_fused_pmc.json) -- not the default
        self.train_rows = True              # training on a GPU: the two convolutions on voxel rows (_fuse_train_rows); False =
                                            # the reference's literal op sequence (cat / reshape / Conv3d modules)
        # Storage type of the inference history ring (T+1 frames per sample): float32 (the reference), or float16 / bfloat16
        # -- BASELINE configs[4] names fp16; at 400x400x16 the ring is 13 GB per sample in fp32.  Sampling and the two
        # convolutions stay fp32 (taps widened exactly, one nearest-even rounding when a frame is stored); the autograd
        # path always keeps fp32.
        self.history_dtype = history_dtype
        # Arithmetic of the two fused inference convolutions: float32 (the reference, fbocc.py:279-282 force_fp32), or
        # bfloat16 = both GEMMs on the bf16 MFMA with fp32 accumulation (weights, frames and the ReLU'd intermediate rounded to
        # bf16; ~4e-3 of the output peak) -- at 400x400x16 the fp32-MFMA kernel is compute bound; or
        # 'bf16x3' = fp32-GRADE results on the bf16 MFMA (every operand split into two bf16 terms, three MFMAs per product:
        # ~5e-6 of the output peak; 16-bit voxel-major ring, C = Cout in {16, 80} -- anything else runs float32); or
        # 'auto' (the DEFAULT since round 4) = 'bf16x3' for a 16-bit ring, float32 for an fp32 ring: with a 16-bit ring the
        # frames already carry a 2^-11 (fp16) / 2^-8 (bf16) storage rounding, against which the split-operand products' 6.5e-6 of
        # the output peak (measured on an MI355X against the fp32 convolutions, tests/test_gpu_history.py) is noise -- and the
        # fp32-MFMA kernel is the compute-bound 10.5 ms of the 14.7 ms configs[4] step (profiles/r03_scope_s6_variants.json).
        # torch.float32 stays selectable and is bit-for-bit the reference's arithmetic on the stored frames.
        self.history_compute = history_compute
        # Layout of the inference ring: 'planar' = the reference's (B, T*C, Z, Y, X); 'voxel_major' = (B, T, N, C) frames of
        # voxel rows (history_kernels.h): a trilinear tap is one 16-byte load of 8 channels instead of 8 scalar gathers from 8
        # planes, and the convolutions read MFMA operands as rows.  Same element bits as the planar ring (the fp32
        # convolutions sum their K in a different order on it: equal to fp32 rounding).  Taken when the register-resident
        # MFMA convolutions are (C = Cout in {16, 80}) -- the DEFAULT since round 3 (measured at BASELINE configs[4]: 21.9 ->
        # 17.2 ms per frame with the reference's arithmetic); other channel counts, and ring_layout='planar', keep the
        # reference's layout; the autograd path stays planar fp32 and either kind of history is converted when the mode changes.  history_bev is then (B, T, N, C): history_as_reference() returns the reference's
        # tensor.
        if ring_layout not in ('planar', 'voxel_major'):
            raise ValueError("ring_layout is 'planar' or 'voxel_major'")
        self.ring_layout = ring_layout
        self.reset()

    def reset(self):
        self.history_sweep_time = None        # (B,T) CPU float tensor
        self.history_bev = None               # (B,T*C,Z,Y,X) GPU
        self.history_seq_ids = None           # (B,) CPU long
        self.history_forward_augs = None      # (B,4,4) GPU
        self._bufs = None
        self._grid = None                     # (Z,Y,X) of the last frame
        self._fold = None                     # cached folded weights (_folded_pair)

    def _voxel_major(self):
        C = self.single_bev_num_channels
        cout = self.history_keyframe_cat_conv[0].weight.shape[0]
        return self.ring_layout == 'voxel_major' and self.use_mfma_convs and C == cout and C in (16, 80)

    def history_as_reference(self):
        """history_bev in the reference's layout and type: (B, T*C, Z, Y, X) fp32 (fbocc.py:234, :312)."""
        h = self.history_bev
        if h is None or h.dim() == 5:
            return None if h is None else h.float()
        B, T, N, C = h.shape
        return h.float().transpose(2, 3).reshape(B, T * C, *self._grid)

    # ------------------------------------------------------------------ pieces
    @staticmethod
    def forward_augs(bda):
        """generate_forward_transformation_matrix (fbocc.py:36-41), no per-sample Python loop."""
        m = torch.zeros((bda.shape[0], 4, 4), dtype=torch.float32, device=bda.device)
        m[:, :3, :3] = bda
        m[:, 3, 3] = 1.0
        return m

    def rt_flow(self, curr_to_prev_ego_rt, bda):
        return _capi.history_flow(self.history_forward_augs, curr_to_prev_ego_rt.contiguous().float(),
                                  bda.contiguous().float(), self.dx, self.lower)

    def _folded(self, seq):
        """1x1x1 conv + eval batch norm -> (weight (Cout,Cin), bias (Cout)) of the equivalent affine map."""
        conv, bn = seq[0], seq[1]
        scale = bn.weight / torch.sqrt(bn.running_var + bn.eps)
        w = conv.weight.flatten(1) * scale[:, None]
        b = (conv.bias - bn.running_mean) * scale + bn.bias
        return w, b

    def _folded_pair(self):
        """Both folded maps, split the way the kernels take them -- (w1 (C,C), time column (C), b1 (C), w2 (Cout,(T+1)C), b2) --
        and cached while no parameter / running statistic changes (storage pointer + in-place version counter of each): the
        dozen small element-wise launches of the folding are a third of the inference step at 100x100x8, B=1."""
        C = self.single_bev_num_channels
        ts = [t for seq in (self.history_keyframe_time_conv, self.history_keyframe_cat_conv)
              for t in (seq[0].weight, seq[0].bias, seq[1].weight, seq[1].bias, seq[1].running_mean, seq[1].running_var)]
        key = tuple((t.data_ptr(), t._version) for t in ts)
        if self._fold is None or self._fold[0] != key:
            with torch.no_grad():
                w1, b1 = self._folded(self.history_keyframe_time_conv)
                w2, b2 = self._folded(self.history_keyframe_cat_conv)
                self._fold = (key, (w1[:, :C].contiguous(), w1[:, C].contiguous(), b1.contiguous(), w2.contiguous(), b2.contiguous()))
        return self._fold[1]

    # ------------------------------------------------------------------ fuse_history (fbocc.py:207-319)
    def fuse_history(self, curr_bev, img_metas, bda):
        if curr_bev.dim() != 5:
            raise NotImplementedError('2-D BEV history (nx[-1] == 1) is outside the built path: FB-OCC fuses a voxel grid')
        _capi.require_gpu(curr_bev, 'curr_bev')
        T, C = self.history_cat_num, self.single_bev_num_channels
        dev = curr_bev.device
        curr = curr_bev.permute(0, 1, 4, 2, 3).float()                 # n, c, z, h, w   (:212)
        B, _, Z, Y, X = curr.shape
        self._grid = (Z, Y, X)
        seq_ids = torch.LongTensor([m['sequence_group_idx'] for m in img_metas])
        start = torch.BoolTensor([bool(m['start_of_sequence']) for m in img_metas])
        fwd = self.forward_augs(bda.float())                           # :220
        ego = torch.stack([torch.as_tensor(m['curr_to_prev_ego_rt'], dtype=torch.float32) for m in img_metas]).to(
            dev, non_blocking=True)                                    # :222-224
        # Like BatchNorm itself, the route is decided by the module's MODE: a module in training mode normalises with batch
        # statistics and updates the running ones even under no_grad (a validation loss computed without .eval());
        # only eval mode without gradients takes the folded-statistics MFMA kernel
        train_path = self.training or (torch.is_grad_enabled() and curr.requires_grad)

        if self.history_bev is None:                                   # first batch (:227-238)
            self.history_bev = self._new_history(curr, train_path)
            self.history_seq_ids = seq_ids.clone()
            self.history_forward_augs = fwd.clone()
            self.history_sweep_time = torch.zeros(B, T)
        self.history_bev = self.history_bev.detach()                   # :241
        bad = (self.history_seq_ids != seq_ids)[~start]
        assert int(bad.sum()) == 0, '{}, {}, {}'.format(self.history_seq_ids, seq_ids, start)   # :248-249
        self.history_sweep_time = self.history_sweep_time + 1          # :252
        if bool(start.any()):                                          # :253-261 (indices known on the host: no sync)
            for b in torch.nonzero(start).flatten().tolist():
                if self.history_bev.dim() == 4:                        # voxel-major ring: (T, N, C) rows
                    self.history_bev[b].copy_(curr[b].detach().reshape(C, -1).t().unsqueeze(0).expand(T, Z * Y * X, C))
                else:
                    self.history_bev[b].view(T, C, Z, Y, X).copy_(curr[b].detach().unsqueeze(0).expand(T, C, Z, Y, X))
                self.history_forward_augs[b] = fwd[b]
            self.history_sweep_time[start] = 0
            self.history_seq_ids[start] = seq_ids[start]

        flow = self.rt_flow(ego, bda)                                  # generate_grid :197-203
        sweep = torch.cat([torch.zeros(B, 1), self.history_sweep_time], dim=1)    # :279-281, B x (1+T)
        if train_path:
            out, feats_cat = self._fuse_train(curr, flow, sweep.to(dev, non_blocking=True))
            h = feats_cat[:, :-C].detach()                             # :312
            self.history_bev = h.clone() if feats_cat.requires_grad else h      # (the row path built feats_cat for this alone)
        else:
            with torch.no_grad():
                if self._voxel_major():
                    out, nxt = self._fuse_infer_vm(curr_bev.detach().float(), curr.detach(), flow, sweep.to(dev, non_blocking=True))
                    self.history_bev = nxt[:, :T]                      # (B, T, N, C) view of the buffer just written
                else:
                    out, nxt = self._fuse_infer(curr.detach(), flow, sweep.to(dev, non_blocking=True))
                    self.history_bev = nxt[:, :T * C]                  # view of the buffer just written: no clone
        self.history_sweep_time = sweep[:, :-1]                        # :313
        self.history_forward_augs = fwd.clone()                        # :314
        if not self.do_history:                                        # :317-318
            self.history_bev = None
        return out.permute(0, 1, 3, 4, 2)                              # (B,Cout,Y,X,Z) view, as :315-319

    forward = fuse_history

    # ------------------------------------------------------------------ internals
    def _frame_buffers(self, like, B, Z, Y, X):
        T, C = self.history_cat_num, self.single_bev_num_channels
        shape = (B, (T + 1) * C, Z, Y, X)
        if (self._bufs is None or tuple(self._bufs[0].shape) != shape or self._bufs[0].device != like.device
                or self._bufs[0].dtype != self.history_dtype):
            self._bufs = [torch.empty(shape, dtype=self.history_dtype, device=like.device) for _ in range(2)]
        return self._bufs

    def _frame_buffers_vm(self, like, B, N):
        T, C = self.history_cat_num, self.single_bev_num_channels
        shape = (B, T + 1, N, C)
        if (self._bufs is None or tuple(self._bufs[0].shape) != shape or self._bufs[0].device != like.device
                or self._bufs[0].dtype != self.history_dtype):
            self._bufs = None                                          # release a planar pair first
            self._bufs = [torch.empty(shape, dtype=self.history_dtype, device=like.device) for _ in range(2)]
        return self._bufs

    def _new_history(self, curr, train_path):
        T = self.history_cat_num
        if train_path:
            return curr.detach().repeat(1, T, 1, 1, 1)                 # :234
        B, C, Z, Y, X = curr.shape
        if self._voxel_major():
            a, _ = self._frame_buffers_vm(curr, B, Z * Y * X)
            a[:, :T].copy_(curr.detach().reshape(B, 1, C, -1).transpose(2, 3).expand(B, T, Z * Y * X, C))
            return a[:, :T]
        a, _ = self._frame_buffers(curr, B, Z, Y, X)
        hist = a[:, :T * C]
        hist.view(B, T, C, Z, Y, X).copy_(curr.detach().unsqueeze(1).expand(B, T, C, Z, Y, X))
        return hist

    def _fuse_train_rows(self, curr, sampled, sweep):
        """The two 1x1x1 convolutions of the training path (:288-310) on VOXEL ROWS: the same maps -- conv + batch norm
        (batch statistics, running buffers updated) + ReLU, twice -- evaluated without the reference's concatenations:
          * the time channel is a per-(sample, frame) scalar, so the 81 -> 80 convolution is the 80 -> 80 one plus
            W[:, 80] * tau_t added to its output before the norm (same sum, no (B, 17, 81, Z, Y, X) copy);
          * the 1360 -> 80 convolution over the frame-concatenated channels is sum_t W2_t y_t, accumulated by T + 1 batched
            GEMMs on the frames' row blocks (no frame <-> voxel transposition of the 435 MB intermediate).
        The round-2 path spent 30 ms of the 303 ms training step in those copies (torch's strided transposing copies run at
        ~150 GB/s on these shapes, profiles/r03_train_step_copy_sites.json).  -> fused (B, Cout, Z, Y, X) view."""
        T, C = self.history_cat_num, self.single_bev_num_channels
        B, _, Z, Y, X = curr.shape
        N = Z * Y * X
        conv1, bn1 = self.history_keyframe_time_conv[0], self.history_keyframe_time_conv[1]
        conv2, bn2 = self.history_keyframe_cat_conv[0], self.history_keyframe_cat_conv[1]
        rows_hist = _capi.transpose_last2(sampled.view(B * T, C, N)).view(B, T, N, C)         # LDS-tiled transpose, no grad
        rows_curr = curr.reshape(B, 1, C, N).transpose(2, 3)                                  # autograd reaches curr from here
        x = torch.cat([rows_curr, rows_hist], dim=1)                                          # (B, T+1, N, C)
        w1 = conv1.weight.flatten(1)                                                          # (C, C + 1)
        tau = (sweep * self.history_cam_sweep_freq).to(x.dtype)                               # (B, T+1)
        y = F.linear(x, w1[:, :C]) + (tau[:, :, None, None] * w1[:, C] + conv1.bias)[:, :, None, :].reshape(B, T + 1, 1, C)
        y = torch.relu_(bn1(y.view(-1, C))).view(B, T + 1, N, C)
        w2 = conv2.weight.flatten(1)                                                          # (Cout, (T+1) C)
        cout = w2.shape[0]
        out = conv2.bias.to(y.dtype).expand(B, N, cout)
        # unbind, not y[:, t]: the backward of T + 1 separate selects materialises a zero-filled full-size tensor per frame
        # and adds them up (17 x 435 MB fills + adds: 18 ms per step); unbind's backward is one stack
        for t, yt in enumerate(y.unbind(1)):                                                  # sum_t y_t W2_t^T
            out = torch.baddbmm(out, yt, w2[:, t * C:(t + 1) * C].t().unsqueeze(0).expand(B, C, cout))
        out = torch.relu_(bn2(out.reshape(-1, cout)))
        return out.view(B, Z, Y, X, cout).permute(0, 4, 1, 2, 3)

    def _fuse_train(self, curr, flow, sweep):
        """The reference's op sequence (:264-310) on this module's layers; the warp is the HIP kernel."""
        T, C = self.history_cat_num, self.single_bev_num_channels
        B, _, Z, Y, X = curr.shape
        hist = self.history_as_reference()                             # the autograd path is planar fp32 (the ring may be neither)
        if hist.stride()[1:] != (Z * Y * X, Y * X, X, 1):
            hist = hist.contiguous()
        sampled = _capi.history_warp(hist, flow, torch.empty((B, T * C, Z, Y, X), dtype=torch.float32, device=curr.device))
        if curr.is_cuda and self.train_rows:
            out = self._fuse_train_rows(curr, sampled, sweep)
            return out, torch.cat([curr.detach(), sampled], dim=1)     # the new history = [current, warped frames] (:286, :312)
        feats_cat = torch.cat([curr, sampled], dim=1)                  # :286
        f = feats_cat.reshape(B, T + 1, C, Z, Y, X)
        tchan = (sweep * self.history_cam_sweep_freq)[:, :, None, None, None, None].expand(B, T + 1, 1, Z, Y, X)
        f = torch.cat([f, tchan], dim=2)                               # :292-295
        f = self.history_keyframe_time_conv(f.reshape(-1, C + 1, Z, Y, X)).reshape(B, T + 1, -1, Z, Y, X)   # :303-305
        out = self.history_keyframe_cat_conv(f.reshape(B, -1, Z, Y, X))                                      # :308-310
        return out, feats_cat

    def _fuse_infer_vm(self, curr_yxz, curr_zyx, flow, sweep):
        """_fuse_infer on the voxel-major ring; curr_yxz is the (B, C, Y, X, Z) volume as handed over, curr_zyx the same
        tensor permuted to (B, C, Z, Y, X) (fbocc.py:212) -- whichever of the two is contiguous feeds the slot-0 transpose
        (the view transformation returns a (Y, X, Z)-shaped VIEW of a (Z, Y, X) buffer: no copy either way)."""
        T, C = self.history_cat_num, self.single_bev_num_channels
        B, _, Y, X, Z = curr_yxz.shape
        n = Z * Y * X
        hist = self.history_bev
        if hist.dim() == 5:                                            # planar history (training path, or the mode changed)
            hist = hist.reshape(B, T, C, n)
            a, b = self._frame_buffers_vm(curr_yxz, B, n)
            a[:, :T].copy_(hist.transpose(2, 3))
            hist, nxt = a[:, :T], b
        else:
            a, b = self._frame_buffers_vm(curr_yxz, B, n)
            if hist.dtype != self.history_dtype:                       # the storage type was changed between frames
                hist = hist.to(self.history_dtype)
            nxt = b if hist.data_ptr() == a.data_ptr() else a
        if curr_zyx.is_contiguous():                                   # slot 0 = current frame (:286)
            _capi.history_frame_vm(curr_zyx.view(B, C, n), nxt[:, 0])
        else:
            _capi.history_frame_vm(curr_yxz.contiguous().view(B, C, n), nxt[:, 0], inner=Z)
        w1, wt, b1, w2, b2 = self._folded_pair()
        tau = (sweep * self.history_cam_sweep_freq).reshape(B * (T + 1), 1)
        bias1 = b1[None, :] + tau * wt[None, :]                        # folded bias + scale * W[:, C] * tau, (B*(T+1), C)
        if (self.fused_warp_conv and self.history_compute == torch.bfloat16 and C == 80 and w2.shape[0] == 80
                and nxt.dtype in (torch.bfloat16, torch.float16) and min(Z, Y, X) >= 2):
            # one launch: warp into slots 1..T and both convolutions from an LDS tile (the T new frames are not read back)
            out = _capi.history_fused_vm(hist, flow, nxt, (Z, Y, X), w1, bias1.contiguous(), w2, b2,
                                         torch.empty((B, 80, n), dtype=torch.float32, device=curr_yxz.device))
            return out.view(B, -1, Z, Y, X), nxt
        compute = self.history_compute
        if compute == 'auto':
            compute = 'bf16x3' if nxt.dtype in (torch.bfloat16, torch.float16) else torch.float32
        if compute == 'bf16x3' and not (nxt.dtype in (torch.bfloat16, torch.float16) and C == w2.shape[0] and C in (16, 80)):
            compute = torch.float32
        if (compute == 'bf16x3' and self.fused_x3 and C == 80 and w2.shape[0] == 80 and min(Z, Y, X) >= 2
                and hist.stride()[1:] == (n * C, C, 1)):
            # ONE launch: every MFMA wave warps its own operands (same ring and volume bits as the two kernels below)
            out = _capi.history_fused_x3_vm(hist, flow, nxt, (Z, Y, X), w1, bias1.contiguous(), w2, b2,
                                            torch.empty((B, 80, n), dtype=torch.float32, device=curr_yxz.device))
            return out.view(B, -1, Z, Y, X), nxt
        if compute == 'bf16x3' and self.pipelined_step and min(Z, Y, X) >= 2 and hist.stride()[1:] == (n * C, C, 1):
            # warp and convolutions band of rows by band of rows on two streams (same kernels, same bits: fbbev_history_step_x3_vm)
            out = _capi.history_step_x3_vm(hist, flow, nxt, (Z, Y, X), w1, bias1.contiguous(), w2, b2,
                                           torch.empty((B, w2.shape[0], n), dtype=torch.float32, device=curr_yxz.device),
                                           chunks=self.pipelined_step_chunks)
            return out.view(B, -1, Z, Y, X), nxt
        _capi.history_warp_vm(hist, flow, nxt[:, 1:], (Z, Y, X))                            # slots 1..T (:275)
        out = _capi.history_conv(nxt, w1, bias1, w2, b2,
                                 torch.empty((B, w2.shape[0], n), dtype=torch.float32, device=curr_yxz.device),
                                 compute=compute, voxel_major=True)
        return out.view(B, -1, Z, Y, X), nxt

    def _fuse_infer(self, curr, flow, sweep):
        T, C = self.history_cat_num, self.single_bev_num_channels
        B, _, Z, Y, X = curr.shape
        n = Z * Y * X
        hist = self.history_bev
        if hist.dim() == 4:                                            # voxel-major history, planar mode now
            hist = self.history_as_reference()
            self._bufs = None
        a, b = self._frame_buffers(curr, B, Z, Y, X)
        nxt = b if hist.data_ptr() == a.data_ptr() else a              # the buffer the history does NOT live in
        if hist.data_ptr() not in (a.data_ptr(), b.data_ptr()):        # history came from the training path
            a[:, :T * C].copy_(hist)
            hist, nxt = a[:, :T * C], b
        nxt[:, :C].copy_(curr)                                         # slot 0 = current frame (:286)
        _capi.history_warp(hist, flow, nxt[:, C:])                     # slots 1..T = aligned history (:275)
        w1c, wt, b1, w2, b2 = self._folded_pair()
        tau = (sweep * self.history_cam_sweep_freq).reshape(B * (T + 1), 1)
        # folded bias already contains scale * conv.bias; the time channel adds scale * W[:, C] * tau = w1[:, C] * tau
        bias1 = b1[None, :] + tau * wt[None, :]                        # (B*(T+1), C)
        cout = w2.shape[0]
        if self.use_mfma_convs and C % 16 == 0 and cout % 16 == 0 and max(C, cout) <= 128:
            # both convs in one MFMA kernel: the (T+1)*C-channel intermediate never leaves the CU
            compute = self.history_compute if (C == cout and C in (16, 80)) else torch.float32
            if compute in ('bf16x3', 'auto'):       # the split-operand kernel reads voxel rows only
                compute = torch.float32
            out = _capi.history_conv(nxt.view(B, (T + 1) * C, n), w1c, bias1.contiguous(), w2, b2,
                                     torch.empty((B, cout, n), dtype=torch.float32, device=curr.device), compute=compute)
        else:
            y = torch.baddbmm(bias1.unsqueeze(-1), w1c.unsqueeze(0).expand(B * (T + 1), C, C),
                              nxt.view(B * (T + 1), C, n).float())
            y.relu_()
            out = torch.baddbmm(b2.view(1, -1, 1), w2.unsqueeze(0).expand(B, *w2.shape), y.view(B, (T + 1) * C, n))
            out.relu_()
        return out.view(B, -1, Z, Y, X), nxt
