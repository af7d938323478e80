import os, sys
os.environ["RL_DEBUG_SYNC"]="1"
sys.path.insert(0,"3d_recognizer_amd"); sys.path.insert(0,".")
import numpy as np, torch
import bench
from randlanet._train import TrainStep
dev=torch.device("cuda",0)
model=bench.build_model(dev); model.train()
st=TrainStep(model,4,40960,use_graph=False)
xyz,labels=bench.synthetic_batch(4,40960,2,1234)
st.set_batch(torch.from_numpy(xyz).to(dev), torch.from_numpy(labels).to(dev))
for i in range(2):
    print("STEP",i,flush=True)
    st.step(np.random.permutation(40960))
torch.cuda.synchronize(); print("done", st.last_metrics())
