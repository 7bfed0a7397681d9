"""Where the per-tensor gradient deviations of the training step come from (the allowances of
tests/test_gpu_train_step.py::judge_grads and __graft_entry__.smoke were justified by an explanation no test isolated).

The loss of config 5 is piecewise smooth: ReLU masks, max-pool selections, the bilinear cells of the deformable sampling, the
arg-max anchors.  An activation that lies within rounding of a kink falls on one side or the other depending on the summation
order, and the gradient of the tensors around it moves by a fraction of a per cent of its norm.  This test measures that effect ON
THE REFERENCE'S OWN ARITHMETIC: the CPU oracle in float64 is the yard-stick, the same oracle in float32 (= the reference on
PyTorch-CPU) deviates from it by up to ~1e-2 of a tensor's gradient norm on a few dozen of the 536 tensors - and the HIP step is
held to that class: no further from the float64 gradient than the float32 reference is (a factor 2 for the different flips two
fp32 evaluations take), tensor by tensor in aggregate and over the whole vector.  Two fp32 evaluations that are each within d of
float64 can differ from each other by 2 d: that is the allowance the batch-2 golden test and smoke() carry."""
import copy
import os

import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _rel(grads, ref, names, floor):
    rows, num, den = [], 0.0, 0.0
    for k in names:
        if ref[k] is None:
            continue
        r = ref[k].double()
        dn = float((grads[k].double().cpu() - r).norm())
        num, den = num + dn * dn, den + float(r.norm()) ** 2
        rows.append((dn / (float(r.norm()) + floor), k))
    return (num / den) ** 0.5, sorted(rows, reverse=True)


def test_hip_gradients_are_as_close_to_float64_as_the_float32_reference_is(calib_dir):
    from egorear_amd import configs, synth, train
    from egorear_amd.estimator import EgoPoseFormerMVFEX
    from oracle import egorear_oracle as O
    from oracle import train_oracle as TO
    torch.set_num_threads(min(16, os.cpu_count() or 1))
    net = EgoPoseFormerMVFEX(**copy.deepcopy(configs.pose3d_cfg("ego4view_rw")))
    sd = synth.load_synth(net, 42)
    names = [k for k, _ in net.named_parameters()]
    B = 2
    img, ctm, gtp, gth = synth.synth_images(B, 4, seed=8), synth.synth_coord_trans_mat(B), synth.synth_gt_pose(B), TO.synth_gt_heatmap(B)
    cams = O.make_cameras("ego4view_rw", calib_dir)
    l32, g32, _, _ = TO.forward_backward({k: v.clone() for k, v in sd.items()}, cams, img, ctm, gtp, gth, names)
    sd64 = {k: (v.double() if v.is_floating_point() else v.clone()) for k, v in sd.items()}
    l64, g64, _, _ = TO.forward_backward(sd64, cams, img.double(), ctm.double(), gtp.double(), gth.double(), names)
    net = net.to(DEV)
    S, _ = train.forward_backward(net, img.to(DEV), ctm.to(DEV), gtp.to(DEV), gth.to(DEV))
    torch.cuda.synchronize()
    for k in names:
        assert (k in S.pgrads) == (g64[k] is not None), k
    floor = 1e-6 * max(float(g.norm()) for g in g64.values() if g is not None)
    glob_cpu, rows_cpu = _rel(g32, g64, names, floor)
    glob_hip, rows_hip = _rel(S.pgrads, g64, names, floor)
    worst_cpu, worst_hip = rows_cpu[0][0], rows_hip[0][0]
    over = lambda rows, t: sum(1 for d, _ in rows if d > t)
    print(f"\nfloat32 reference vs float64: whole vector {glob_cpu:.2e}, worst tensor {worst_cpu:.2e} ({rows_cpu[0][1]}), "
          f"{over(rows_cpu, 1e-3)} tensors above 1e-3, {over(rows_cpu, 1e-4)} above 1e-4 of {len(rows_cpu)}")
    print(f"HIP step          vs float64: whole vector {glob_hip:.2e}, worst tensor {worst_hip:.2e} ({rows_hip[0][1]}), "
          f"{over(rows_hip, 1e-3)} tensors above 1e-3, {over(rows_hip, 1e-4)} above 1e-4")
    # the effect exists in the reference's own arithmetic (otherwise this test pins nothing) ...
    assert worst_cpu > 1e-3 and over(rows_cpu, 1e-3) >= 3
    # ... and the HIP step is in the same class: loss, whole vector, worst tensor, number of affected tensors
    assert abs(float(S.loss_terms.sum()) - sum(l64.values())) <= 1e-5 * sum(l64.values())
    assert glob_hip <= 2.0 * glob_cpu + 2e-5, (glob_hip, glob_cpu)
    assert worst_hip <= 2.0 * worst_cpu + 1e-3, (rows_hip[:4], rows_cpu[:4])
    assert over(rows_hip, 1e-3) <= 2 * over(rows_cpu, 1e-3) + 5
    assert over(rows_hip, 1e-4) <= 2 * over(rows_cpu, 1e-4) + 20
    # the bulk of the tensors (no kink nearby) agrees with float64 to rounding
    med = sorted(d for d, _ in rows_hip)[len(rows_hip) // 2]
    assert med <= 2e-5, med
