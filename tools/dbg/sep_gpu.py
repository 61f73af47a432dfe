import sys, numpy as np, torch
sys.path.insert(0, 'tests')
import _cases as C
from helpers import build_separable_case, strict_share, box_iou
from oracle import models as om, nms as onms
from oracle.policy import run_policy
from pytorch_yolo_amd.utils.utils import non_max_suppression
sep = C.SEPARABLE
model, sd, x, g = build_separable_case()
ref = g["nms_dets_0"]; kept = g["nms_kept_0"]
model = model.to("cuda:0")
with torch.no_grad():
    io16, p16 = model(x.to("cuda:0"))
    dets16, idx16 = non_max_suppression(io16, sep["conf_thres"], sep["nms_thres"], with_indices=True)
d16 = dets16[0].cpu().numpy(); k16 = idx16[0].cpu().numpy()
io16 = io16[0].cpu().numpy()
with torch.no_grad():
    io_f, _ = om.spp_forward(sd, x, C.SPP_ANCHORS, 80)
io_b, _ = run_policy(om.spp_forward, sd, x, C.SPP_ANCHORS, 80, policy="bf16")
io_f, io_b = io_f.numpy()[0], io_b.numpy()[0]
def desc(rows, other, tag):
    for r in rows:
        best = max(other, key=lambda q: box_iou(r[:4], q[:4]))
        print(f"  {tag}: cls {int(r[6])} conf {r[4]:.3f} box {np.round(r[:4],1)} | best other: cls {int(best[6])} conf {best[4]:.3f} iou {box_iou(r[:4], best[:4]):.3f}")
un_ref = [r for r in ref if not any(int(r[6])==int(q[6]) and abs(r[4]-q[4])<=0.03 and box_iou(r[:4],q[:4])>=0.9 for q in d16)]
un_hip = [r for r in d16 if not any(int(r[6])==int(q[6]) and abs(r[4]-q[4])<=0.03 and box_iou(r[:4],q[:4])>=0.9 for q in ref)]
desc(un_ref, d16, "ref w/o partner"); desc(un_hip, ref, "hip w/o partner")
print("kept only in ref:", sorted(set(kept.tolist())-set(k16.tolist())), "only in hip:", sorted(set(k16.tolist())-set(kept.tolist())))
for row in sorted(set(kept.tolist()) ^ set(k16.tolist())):
    print(f"  row {row}: obj fp32 {io_f[row,4]:.4f} policy {io_b[row,4]:.4f} hip {io16[row,4]:.4f}; clsmax fp32 {io_f[row,5:].max():.4f} policy {io_b[row,5:].max():.4f} hip {io16[row,5:].max():.4f}")
# how far is hip from policy vs each from fp32 on the obj channel, in logit units of the candidates
cand = np.nonzero((io_f[:,4]*io_f[:,5:].max(1) > 0.05) | (io16[:,4]*io16[:,5:].max(1) > 0.05))[0]
lg = lambda p_: np.log(np.clip(p_,1e-30,1)/np.clip(1-p_,1e-30,1))
print("candidates", len(cand), "obj logit |hip - policy| median %.3f max %.3f; |policy - fp32| median %.3f max %.3f; |hip - fp32| median %.3f max %.3f" % (
    np.median(np.abs(lg(io16[cand,4])-lg(io_b[cand,4]))), np.abs(lg(io16[cand,4])-lg(io_b[cand,4])).max(),
    np.median(np.abs(lg(io_b[cand,4])-lg(io_f[cand,4]))), np.abs(lg(io_b[cand,4])-lg(io_f[cand,4])).max(),
    np.median(np.abs(lg(io16[cand,4])-lg(io_f[cand,4]))), np.abs(lg(io16[cand,4])-lg(io_f[cand,4])).max()))
