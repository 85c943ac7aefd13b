// Host-side launch plans (round 3): whole kernel sequences of the hot path issued by ONE C call.
//
// The eager training step on real, changing image shapes (detectron2/data/dataset_mapper.py:112-185 gives another
// (H, W, R) every iteration, so no hipGraph can be replayed) was bound by the HOST: ~150 launches per step, each a Python
// module call -> ops wrapper -> ctypes marshalling of 20-25 arguments -> HIP launch, 1.5-1.7 ms of enqueue per step
// against ~1.5 ms of GPU work on the main stream (profiles/r3_10_eager800_*).  The frozen trunk is a third of those launches
// (WS-ResNet50-C4: 46 convolutions + 3 max-pools; resnet_ws.py:405-416, :217-237, :479-502; vgg.py:104-122).  Its layer
// sequence never changes, only the image size does - so the sequence is recorded ONCE as an array of DrnTrunkOp (weights,
// folded FrozenBN affine, geometry, activation-slot indices) and every forward is one call: shapes per layer are derived
// here from (Nb, H, W), and each op goes through the SAME entry points the per-layer path uses (drn_conv2d_nhwc_q,
// drn_maxpool2x2_nhwc: identical kernel selection, identical launches, bit-identical results).
#include "drn_common.h"
#include "../../include/drn_wsod.h"

namespace {

struct SlotGeom { int h, w, c, es; bool set; };

// walks the plan, calling `visit(op, in geometry, out geometry)`; returns DRN_OK or an argument error
template <class F>
int walk(const DrnTrunkOp* ops, int n_ops, int n_slots, int in_slot, int H, int W, int C0, int in_dtype, SlotGeom* g, F&& visit) {
  if (!ops || n_ops < 0 || n_slots <= 0 || n_slots > DRN_TRUNK_MAX_SLOTS || in_slot < 0 || in_slot >= n_slots) return DRN_ERR_ARG;
  for (int s = 0; s < n_slots; ++s) g[s].set = false;
  g[in_slot] = SlotGeom{H, W, C0, drn_esize(in_dtype), true};
  for (int i = 0; i < n_ops; ++i) {
    const DrnTrunkOp& o = ops[i];
    if (o.src < 0 || o.src >= n_slots || o.dst < 0 || o.dst >= n_slots || o.dst == o.src || !g[o.src].set) return DRN_ERR_ARG;
    const SlotGeom in = g[o.src];
    SlotGeom out;
    const int kind = o.kind & DRN_TRUNK_KIND_MASK;
    if (kind == DRN_TRUNK_CONV) {
      if (in.c != o.cin || in.es != drn_esize(o.dtype)) return DRN_ERR_ARG;
      const int ho = (in.h + 2 * o.pad - o.dil * (o.ksize - 1) - 1) / o.stride + 1;
      const int wo = (in.w + 2 * o.pad - o.dil * (o.ksize - 1) - 1) / o.stride + 1;
      if (ho <= 0 || wo <= 0) return DRN_ERR_ARG;
      out = SlotGeom{ho, wo, o.cout, drn_esize(o.out_dtype), true};
      if (o.res >= 0) {
        if (o.res >= n_slots || o.res == o.dst || !g[o.res].set) return DRN_ERR_ARG;
        const SlotGeom& r = g[o.res];
        if (r.h != ho || r.w != wo || r.c != o.cout || r.es != drn_esize(o.res_dtype)) return DRN_ERR_ARG;
      }
    } else if (kind == DRN_TRUNK_MAXPOOL) {
      if (in.h < 2 || in.w < 2 || (o.stride != 1 && o.stride != 2)) return DRN_ERR_ARG;
      out = SlotGeom{(in.h - 2) / o.stride + 1, (in.w - 2) / o.stride + 1, in.c, in.es, true};
    } else {
      return DRN_ERR_ARG;
    }
    const int rc = visit(o, in, out);
    if (rc != DRN_OK) return rc;
    g[o.dst] = out;
  }
  return DRN_OK;
}

// op `pl` is the 2x2 / stride-2 max pool of conv op `c`'s output, into a slot that exists
static bool pool_follows(const DrnTrunkOp& pl, const DrnTrunkOp& c, int n_slots, void* const* slots) {
  return (pl.kind & DRN_TRUNK_KIND_MASK) == DRN_TRUNK_MAXPOOL && pl.stride == 2 && pl.src == c.dst && pl.dst >= 0 &&
         pl.dst < n_slots && slots[pl.dst] && c.relu;
}

}  // namespace

extern "C" {

int drn_trunk_shapes(const DrnTrunkOp* ops, int n_ops, int n_slots, int in_slot, int Nb, int H, int W, int C0,
                     int in_dtype, long* slot_bytes, int* slot_hwc) {
  if (!slot_bytes || Nb <= 0 || H <= 0 || W <= 0) return DRN_ERR_ARG;
  SlotGeom g[DRN_TRUNK_MAX_SLOTS];
  for (int s = 0; s < n_slots && s < DRN_TRUNK_MAX_SLOTS; ++s) slot_bytes[s] = 0;
  const int rc = walk(ops, n_ops, n_slots, in_slot, H, W, C0, in_dtype, g, [&](const DrnTrunkOp& o, const SlotGeom&, const SlotGeom& out) {
    const long b = (long)Nb * out.h * out.w * out.c * out.es;
    if (b > slot_bytes[o.dst]) slot_bytes[o.dst] = b;
    return DRN_OK;
  });
  if (rc != DRN_OK) return rc;
  if (slot_hwc)  // geometry each slot holds when the plan ends (the callers' output views)
    for (int s = 0; s < n_slots; ++s) {
      slot_hwc[3 * s] = g[s].set ? g[s].h : 0;
      slot_hwc[3 * s + 1] = g[s].set ? g[s].w : 0;
      slot_hwc[3 * s + 2] = g[s].set ? g[s].c : 0;
    }
  return DRN_OK;
}

int drn_trunk_forward(const DrnTrunkOp* ops, int n_ops, int n_slots, int in_slot, void* const* slots, int Nb, int H, int W,
                      int C0, int in_dtype, void* stream) {
  if (!slots || Nb <= 0 || H <= 0 || W <= 0) return DRN_ERR_ARG;
  SlotGeom g[DRN_TRUNK_MAX_SLOTS];
  int fused_tail = -1, fused_pool = -1;  // ops that already ran inside an earlier op's launch (the 1x1 tail, the max pool)
  return walk(ops, n_ops, n_slots, in_slot, H, W, C0, in_dtype, g, [&](const DrnTrunkOp& o, const SlotGeom& in, const SlotGeom&) {
    if (!slots[o.src] || !slots[o.dst] || (o.res >= 0 && !slots[o.res])) return DRN_ERR_ARG;
    const int idx = (int)(&o - ops);
    if (idx == fused_tail || idx == fused_pool) return DRN_OK;
    if ((o.kind & DRN_TRUNK_FUSE_NEXT) && idx + 1 < n_ops) {
      // 3x3 (64 -> 64) whose output only the next op - a 1x1 to 256 channels - reads: one launch on large maps
      // (drn_conv3x3_pw_nhwc; bit-identical to the two), the two launches wherever that kernel does not apply
      const DrnTrunkOp& n = ops[idx + 1];
      const bool ok = (n.kind & DRN_TRUNK_KIND_MASK) == DRN_TRUNK_CONV && n.src == o.dst && o.res < 0 && o.ksize == 3 &&
                      o.cin == 64 && o.cout == 64 && o.stride == 1 && o.pad == 1 && o.dil == 1 && n.ksize == 1 && n.cin == 64 &&
                      n.cout == 256 && n.stride == 1 && n.pad == 0 && o.dtype == DRN_BF16 && o.out_dtype == DRN_BF16 &&
                      n.dtype == DRN_BF16 && n.out_dtype == DRN_BF16 && (n.res < 0 || n.res_dtype == DRN_BF16) &&
                      n.dst >= 0 && n.dst < n_slots && slots[n.dst] && (n.res < 0 || (n.res < n_slots && slots[n.res]));
      if (ok) {
        // ... and the max pool behind the 1x1, when that op is flagged too (the last block of res2): three ops, one launch
        const bool pool = (n.kind & DRN_TRUNK_FUSE_POOL) && idx + 2 < n_ops && pool_follows(ops[idx + 2], n, n_slots, slots);
        if (pool) {
          const int rc = drn_conv3x3_pw_nhwc(slots[o.src], o.w, o.scale, o.bias, o.relu, n.w, n.scale, n.bias,
                                             n.res >= 0 ? slots[n.res] : nullptr, slots[ops[idx + 2].dst], Nb, in.h, in.w, o.ldw,
                                             n.ldw, n.res_mult, n.relu, 1, stream);
          if (rc == DRN_OK) { fused_tail = idx + 1; fused_pool = idx + 2; return DRN_OK; }
          if (rc != DRN_ERR_UNSUPPORTED) return rc;
        }
        const int rc = drn_conv3x3_pw_nhwc(slots[o.src], o.w, o.scale, o.bias, o.relu, n.w, n.scale, n.bias,
                                           n.res >= 0 ? slots[n.res] : nullptr, slots[n.dst], Nb, in.h, in.w, o.ldw, n.ldw,
                                           n.res_mult, n.relu, 0, stream);
        if (rc == DRN_OK) { fused_tail = idx + 1; return DRN_OK; }
        if (rc != DRN_ERR_UNSUPPORTED) return rc;
      }
    }
    if ((o.kind & DRN_TRUNK_FUSE_POOL) && !(o.kind & DRN_TRUNK_FUSE_NEXT) && idx + 1 < n_ops &&
        (o.kind & DRN_TRUNK_KIND_MASK) == DRN_TRUNK_CONV && o.ksize == 3 && o.cin == 64 && o.cout == 64 && o.stride == 1 &&
        o.pad == 1 && o.dil == 1 && o.relu && o.dtype == DRN_BF16 && o.out_dtype == DRN_BF16 &&
        (o.res < 0 || o.res_dtype == DRN_BF16) && pool_follows(ops[idx + 1], o, n_slots, slots)) {
      // a 3x3 / 64 -> 64 conv with the max pool behind it (the deep stem's last conv): the pooled map is all that is written
      const int rc = drn_conv3x3_pw_nhwc(slots[o.src], o.w, o.scale, o.bias, o.relu, nullptr, nullptr, nullptr,
                                         o.res >= 0 ? slots[o.res] : nullptr, slots[ops[idx + 1].dst], Nb, in.h, in.w, o.ldw, 0,
                                         o.res_mult, 0, 1, stream);
      if (rc == DRN_OK) { fused_pool = idx + 1; return DRN_OK; }
      if (rc != DRN_ERR_UNSUPPORTED) return rc;
    }
    if ((o.kind & DRN_TRUNK_KIND_MASK) == DRN_TRUNK_CONV)
      return drn_conv2d_nhwc_q(slots[o.src], o.w, slots[o.dst], o.scale, o.bias, o.res >= 0 ? slots[o.res] : nullptr, Nb, in.h,
                               in.w, o.cin, o.cout, o.ksize, o.ksize, o.stride, o.pad, o.dil, o.ldw, o.cout, o.cout, o.relu,
                               o.dtype, o.out_dtype, o.res >= 0 ? o.res_dtype : o.out_dtype, o.res_mult, stream);
    return drn_maxpool2x2_nhwc(slots[o.src], slots[o.dst], Nb, in.h, in.w, in.c, o.stride, o.dtype, stream);
  });
}

}  // extern "C"
