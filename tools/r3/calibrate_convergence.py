import sys, time; sys.path.insert(0,'/root/repo')
import path_space_pde_solver_amd as psp, torch, numpy as np
torch.set_num_threads(4)
def err(m, pb):
    g=torch.Generator().manual_seed(5); xp = 0.5*torch.randn(16, pb.d, generator=g)
    out=[]
    for t in (0.0, 0.2, 0.4):
        with torch.no_grad(): u = -m.Z_n(xp, t)
        ut = torch.tensor(np.asarray(pb.u_true(xp, t))).float().t()
        out.append(float((u-ut).norm()/ut.norm()))
    return out
for rx in (True,):
    pb = psp.LQGC(d=4, off_diag=0.1, T=0.5, seed=42, delta_t=0.01, device='cpu')
    m = psp.Solver('conv', pb, lr=0.01, L=2500, K=512, delta_t=0.01, loss_method='log-variance', time_approx='inner', adaptive_forward_process=True,
                   detach_forward=True, u_l2_error_flag=False, verbose=False, seed=42, device='cpu', backend='torch', widths=(30,30), random_X_0=rx)
    e0=err(m,pb); t=time.time(); m.train(); print('lqgc', rx, round(time.time()-t,1), e0, err(m,pb), m.loss_log[0], m.loss_log[-1], flush=True)
pb = psp.HeatEquation(d=4, T=0.5, device='cpu')
m = psp.GeneralSolver(pb, 'heat', seed=42, delta_t=0.01, N=50, lr=0.01, L=1500, K=512, K_boundary=64, loss_method='diffusion', verbose=False, device='cpu', backend='torch')
g=torch.Generator().manual_seed(5); xp = torch.randn(64,4,generator=g); xp = xp/ xp.norm(dim=1,keepdim=True)*torch.rand(64,1,generator=g)
tp = torch.full((64,1),0.25)
def ev():
    with torch.no_grad(): v=m.V(torch.cat([xp,tp],1)).squeeze()
    vt = pb.v_true(xp, tp.squeeze())
    return float((v-vt).norm()/vt.norm())
e0=ev(); t=time.time(); m.train(); print('heat', round(time.time()-t,1), e0, ev(), m.loss_log[0], m.loss_log[-1], flush=True)
pb = psp.ExponentialOnSphereNonlinearParabolic(d=4, T=0.5, alpha=0.5, device='cpu')
for L in (400, 1200):
    m = psp.GeneralSolver(pb, 'exps', seed=42, delta_t=0.01, N=50, lr=0.01, L=L, K=512, K_boundary=64, loss_method='diffusion', verbose=False, device='cpu', backend='torch')
    g=torch.Generator().manual_seed(5); xp = torch.randn(64,4,generator=g); xp = xp/ xp.norm(dim=1,keepdim=True)*torch.rand(64,1,generator=g)**0.25
    tp = torch.full((64,1),0.25)
    def ev():
        with torch.no_grad(): v=m.V(torch.cat([xp,tp],1)).squeeze()
        vt = pb.v_true(xp, tp.squeeze())
        return float((v-vt).norm()/vt.norm())
    e0=ev(); t=time.time(); m.train(); print('expsphere', L, round(time.time()-t,1), e0, ev(), m.loss_log[0], m.loss_log[-1], flush=True)
