import os, sys, time, ctypes as C
os.environ["WRENC_GPU_LIB"]=os.path.abspath("scratch/libwrenc_gpu_prof.so")
sys.path.insert(0,'.')
from wrenc_amd import gpu, synth
w,h,qp,depth=1920,1088,32,int(sys.argv[1]) if len(sys.argv)>1 else 2
B=int(sys.argv[2]) if len(sys.argv)>2 else 16
frames=[synth.synth_frame(w,h,f) for f in range(4)]
enc=gpu.Encoder(w,h,qp=qp,max_split_depth=depth,n_slots=B)
for s in range(B):
    enc.upload(s,*frames[s%4])
enc.sync()
names=["predict","fdct","q_pre","q_back","q_trace","deq","idct","recon","total","ctrl","refs","skip","nstep","nfull"]+["sad_t%d_c%d"%(4<<(i//2),i%2) for i in range(8)]+["x"]+["sad_n%d_c%d"%(4<<(i//2),i%2) for i in range(8)]+["x2","qb_pre","qb_wait1","qb_walk","qb_wait2"]
out=(C.c_ulonglong*36)()
enc.lib.wrenc_gpu_prof_read.argtypes=[C.c_void_p, C.c_void_p, C.c_int]
enc.lib.wrenc_gpu_prof_read(enc.ctx, out, 36)
t0=time.time(); enc.encode(0,B); enc.sync(); dt=time.time()-t0
enc.lib.wrenc_gpu_prof_read(enc.ctx, out, 36)
tot=out[8]
nctu=B*2040
print("wall %.3f s, fps %.2f, cycles/CTU total %.0f (100MHz ticks?)"%(dt,B/dt,tot/nctu))
acc=0
for i,n in enumerate(names):
    if i==8: continue
    print("%-8s %6.2f%%  %.0f per CTU"%(n,100.0*out[i]/tot,out[i]/nctu)); acc+=out[i]
print("other    %6.2f%%"%(100.0*(tot-acc)/tot))
