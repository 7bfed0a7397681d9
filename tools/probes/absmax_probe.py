import copy, os, sys
sys.path.insert(0, os.getcwd())
import torch
from egorear_amd import configs, hip, synth, train
from egorear_amd.estimator import EgoPoseFormerMVFEX
from egorear_amd.metrics import generate_target
net = EgoPoseFormerMVFEX(**copy.deepcopy(configs.pose3d_cfg("ego4view_rw"))); synth.load_synth(net, 42); net = net.to("cuda:0")
tr = train.Trainer(net, use_graph=False)
B = 32
img = synth.synth_images(B, 4, seed=1234).to("cuda:0"); ctm = synth.synth_coord_trans_mat(B).to("cuda:0")
gt_pose = synth.synth_gt_pose(B).to("cuda:0"); gt_hm = generate_target(synth.synth_joint_px(B).to("cuda:0")).contiguous()
tr.step(img, ctm, gt_pose, gt_hm)
import traceback
orig = hip.absmax_record
log = []
def probe(t, rec):
    st = [f"{f.name}:{f.lineno}" for f in traceback.extract_stack()[-6:-1]]
    log.append((tuple(t.shape), t.numel() * 4 / 1e6, " < ".join(st)))
    return orig(t, rec)
hip.absmax_record = probe
tr.step(img, ctm, gt_pose, gt_hm)
torch.cuda.synchronize()
for s, mb, st in log: print(s, f"{mb:.1f} MB", st)
