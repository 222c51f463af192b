import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo")); sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "src"))
import torch, numpy as np
from gcn_grabcut import GCNGrabCutPipeline, SuperpixelGraphConfig, ResGCNNet
from gcn_grabcut.synthetic import synthetic_batch
torch.manual_seed(0)
model = ResGCNNet().eval()
pipe = GCNGrabCutPipeline(model, sp_config=SuperpixelGraphConfig(n_segments=600), grabcut_lanes=int(os.environ.get("LANES", "4")))
imgs = synthetic_batch(int(os.environ.get("MF_BATCH", "256")), 300, 400, 3)
out = pipe.segment_batch_device(torch.from_numpy(imgs).cuda())
torch.cuda.synchronize()
