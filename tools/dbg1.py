import sys, numpy as np, torch
sys.path.insert(0,'.')
import opencorr_amd, oracle
from opencorr_amd import synth
dev=torch.device('cuda',0)
side=1024
ref,tar=synth.speckle_pair_2d(side,side,seed=20260925,device=dev)
refh,tarh=ref.cpu().numpy(),tar.cpu().numpy()
ref2,tar2=synth.speckle_pair_2d(side,side,seed=20260925)
print('gpu-vs-cpu synth diff', np.abs(refh-ref2).max(), np.abs(tarh-tar2).max(), 'ref-tar diff', np.abs(refh-tarh).mean())
xs,ys=synth.poi_grid_2d(side,side,40,40,24)
p=opencorr_amd.make_pois2d(xs,ys)
f=opencorr_amd.FFTCC2D(16,16); f.set_images(refh,tarh); f.compute(p)
po=oracle.make_pois2d(xs,ys); oracle.fftcc2d(refh,tarh,16,16,po)
print('fftcc equal', np.array_equal(p[:,[2,8]],po[:,[2,8]]), p[:3,[2,8,16]], po[:3,[2,8,16]])
i=opencorr_amd.ICGN2D1(16,16,0.001,10); i.share_images(f); i.prepare(); g=p.copy(); i.compute(g)
prep=oracle.Prepared2D(refh,tarh); w=p.copy(); oracle.icgn2d1(prep,16,16,0.001,10,w,order=1)
print('icgn bit equal', np.array_equal(g.view(np.uint32),w.view(np.uint32)), 'iters gpu',g[:,17].mean(),'oracle',w[:,17].mean())
eu,ev=synth.expected_deformation_2d(xs,ys,side,side)
print('err', np.abs(w[:,2]-eu).max(), np.abs(w[:,8]-ev).max())
