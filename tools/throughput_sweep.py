import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from wrenc_amd import gpu, synth
for (w,h,qp,depth,B) in [(1920,1088,32,3,256),(3840,2176,32,3,128),(3840,2176,22,3,128),(1920,1088,32,0,512),(1920,1088,32,1,512)]:
    frames=[synth.synth_frame(w,h,f) for f in range(2)]
    enc=gpu.Encoder(w,h,qp=qp,max_split_depth=depth,n_slots=B)
    for s in range(B): enc.upload(s,*frames[s%2])
    enc.sync()
    t0=time.time(); enc.encode(0,B); enc.sync(); dt=time.time()-t0
    print(w,h,'qp',qp,'depth',depth,'B',B,'wall %.3fs'%dt,'fps %.2f'%(B/dt),'Mpix/s %.1f'%(B*w*h/dt/1e6), 'mismatch', enc.final_pass_mismatches(), flush=True)
    enc.close()
