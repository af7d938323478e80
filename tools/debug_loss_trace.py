import os, sys
sys.path.insert(0,"3d_recognizer_amd"); sys.path.insert(0,".")
import numpy as np, torch
import bench
from randlanet._train import TrainStep
dev=torch.device("cuda",0)
use_graph = os.environ.get("GRAPH","1")=="1"
model=bench.build_model(dev); model.train()
st=TrainStep(model,4,40960,use_graph=use_graph)
xyz,labels=bench.synthetic_batch(4,40960,2,1234)
st.set_batch(torch.from_numpy(xyz).to(dev), torch.from_numpy(labels).to(dev))
np.random.seed(1234)
st.capture()
for i in range(45):
    st.step(np.random.permutation(40960))
    m=st.last_metrics()
    pn=float(st.flat.param.abs().max()); gn=float(st.flat.grad.abs().max())
    print(i, "loss %.5f mIoU %.4f  |p|max %.3e |g|max %.3e" % (m["loss"], m["mIoU"], pn, gn), flush=True)
    if not np.isfinite(m["loss"]): break
