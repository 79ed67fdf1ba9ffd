import sys, time, os, numpy as np, torch
sys.path.insert(0,'oracle'); import imf_oracle as O, imf_oracle_cbind as OC
z=np.load('tests/golden/fixture_clouds.npz'); xyz=z['cloud_bin_0'].astype(np.float64)*1.7
img=np.transpose(np.load('tests/golden/fixture_images.npz')['image_0'],(2,0,1))[None].copy()
sd=O.seeded_state_dict(0)
for nt in (int(sys.argv[1]),):
    torch.set_num_threads(nt)
    for r in range(3):
        t=time.time(); c,i=OC.voxelize(xyz,0.025); t1=time.time(); g=OC.Geometry(c); t2=time.time()
        F=O.resunet_forward(sd,c,img,geometry=g); t3=time.time()
        print(nt, os.environ.get('OMP_NUM_THREADS'), 'vox %.3f geom %.3f fwd %.3f total %.3f'%(t1-t,t2-t1,t3-t2,t3-t), flush=True)
