import sys,os,numpy as np,torch
sys.path.insert(0,os.getcwd())
from pcrcg_amd import indoor_config, synthetic
from pcrcg_amd.architectures import KPFCNN
from pcrcg_amd.pyramid import build_pyramid
dev=torch.device("cuda:0"); cfg=indoor_config(); torch.manual_seed(0)
net=KPFCNN(cfg).to(dev).eval()
a,b=synthetic.pair("S30k",100)
pts=torch.from_numpy(np.concatenate([a,b])).to(dev); lens=torch.tensor([len(a),len(b)],dtype=torch.int32,device=dev)
batch=build_pyramid(pts,lens,cfg,synthetic.LIMITS["S30k"])
with torch.no_grad(): net(batch)
torch.cuda.synchronize()
