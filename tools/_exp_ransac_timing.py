import shutil, numpy as np, sys
sys.path.insert(0,'.')
shutil.copy('visual_odometry_amd/libvo_hip_timing.so','visual_odometry_amd/libvo_hip.so')
from visual_odometry_amd import synth
from visual_odometry_amd.frontend import FrontEnd
seq=synth.sequence(9,1280,720,cache_dir='/tmp')
fe=FrontEnd(720,1280,9,8,nfeatures=2000)
fe.upload(seq['frames']); fe.detect(0,9)
pairs=[[i,i+1] for i in range(8)]
opts=fe.make_opts(want_points=False)
res,_=fe.run_pairs(pairs,seq['K'],opts)
for r in res: print("subset",r['n_kp1'],"solve",r['n_kp2'],"score",r['n_match'],"iters",r['reserved'], "inl", r['n_inl'])
