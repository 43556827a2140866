import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, numpy as np
from pcrcg_amd import indoor_config
from pcrcg_amd.architectures import KPFCNN
from pcrcg_amd import _lib
net = KPFCNN(indoor_config()).cuda().eval()
d = net.runner().descriptor()
print("heads", d.heads, "gnn_dim", d.gnn_dim, "n_gnn", d.n_gnn, [d.gnn[i].cross for i in range(d.n_gnn)])
print(_lib.lib().pcrcg_attention_supported(64))
