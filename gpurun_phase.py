import numpy as np, torch, time, sys
sys.path.insert(0,'.')
from tests.problems import C2, c2_x0, product_nmpc
B=1024
nmpc=product_nmpc(C2)
x=torch.as_tensor(c2_x0(B),device='cuda'); p=torch.as_tensor(np.array(C2['p']),device='cuda')
for _ in range(3):
    u=nmpc.optimize(x,cp=p); x=nmpc.plant_step(x,u,cp=p)
nmpc.phase_profile(True)
u=nmpc.optimize(x,cp=p); torch.cuda.synchronize()
pr=nmpc.phase_profile(True)
it=int(nmpc._nlp_solution['iter_count'][0])
print('iters',it,{k:(v if k.startswith('n_') else round(v/it)) for k,v in pr.items()})
