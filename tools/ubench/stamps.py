import os, sys, torch, numpy as np
sys.path.insert(0, "/root/repo")
import textgan_pytorch_amd as tg
V,L,B,R=5255,51,64,16
dev="cuda"
gen=tg.modules.synthetic_generator(V,32,32,L,device=dev)
sent=torch.from_numpy(tg.synth.tokens(1,(B,L),V)).to(dev)
u=torch.rand(R,L,L,B,device=dev)
grid=(R*B//32)*L
st=torch.zeros(grid,8,dtype=torch.int64,device=dev)
tg.ops.gen_rollout(gen,sent,R,u)
os.environ["TG_STAMPS"]=str(st.data_ptr())
tg.ops.gen_rollout(gen,sent,R,u); torch.cuda.synchronize()
s=st.cpu().numpy().astype(np.float64)
n=s[:,3]; m=n>0
print("WGs with steps",m.sum(),"total steps",n.sum())
for i,name in [(0,"cell"),(1,"proj"),(2,"draw"),(4,"draw:barrier0"),(5,"draw:spansum"),(6,"draw:select"),(7,"draw:recompute")]:
    print(name,"cycles/step avg %.0f"%(s[m,i].sum()/n.sum()), " first WG %.0f"%(s[0,i]/n[0]), " last-g WG %.0f"%(s[m][-1,i]/n[m][-1]))
