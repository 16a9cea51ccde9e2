import numpy as np, torch, time, sys
sys.path.insert(0,'.')
from tests.problems import C2, c2_x0, product_nmpc
nmpc=product_nmpc(C2)
B=1024
x=torch.as_tensor(c2_x0(B),device='cuda'); p=torch.as_tensor(np.array(C2['p']),device='cuda')
for _ in range(5):
    u=nmpc.optimize(x,cp=p); x=nmpc.plant_step(x,u,cp=p)
nmpc.phase_profile(True)
for rep in range(3):
    u=nmpc.optimize(x,cp=p); torch.cuda.synchronize()
    pr=nmpc.phase_profile(True)
    it=int(nmpc._nlp_solution['iter_count'][0]); its=nmpc._nlp_solution['iter_count'].cpu().numpy()
    print('iters inst0',it,'max',its.max(),'mean',its.mean(), {k:(v if k.startswith('n_') else round(v/max(it,1))) for k,v in pr.items()})
    x=nmpc.plant_step(x,u,cp=p)
