# PMC FETCH_SIZE of istft_frames_kernel (spectro2wav and the STFT adjoint of the MR-STFT loss) — dev tool
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cat > /tmp/istft_run.py <<'PY'
import sys, torch
sys.path.insert(0, sys.argv[1])
from vm_asr_amd import stft as S
B,F,M=4,513,512
mag=torch.randn(B,1,F,M,device='cuda'); ph=torch.randn(B,1,F,M,device='cuda')
y=S.spectro2wav(mag,ph,1024,240,1024,'log2')
for n,hop,win in ((1024,120,600),(2048,240,1200),(512,50,240)):
    x=torch.randn(4,122640,device='cuda',requires_grad=True)
    re,im=S.stft_reim(x,n,hop,win,False) if hasattr(S,'stft_reim') else (None,None)
    (re.sum()+im.sum()).backward()
torch.cuda.synchronize()
PY
for v in old new; do
cp $R/tools/tmp/$v.so $R/vm_asr_amd/libvmasr_hip.so
rm -rf /tmp/pf
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d /tmp/pf -o f -- python /tmp/istft_run.py $R > /tmp/run_$v.log 2>&1
tail -n 2 /tmp/run_$v.log | cut -c1-200
python - $v <<'PY'
import csv,glob,sys
f=glob.glob('/tmp/pf/**/*counter_collection.csv',recursive=True)
if not f: print(sys.argv[1], 'no counter file'); raise SystemExit
rows=[r for r in csv.DictReader(open(f[0])) if 'istft_frames' in r['Kernel_Name']]
print(sys.argv[1], len(rows), [round(float(r['Counter_Value'])*2*1024/1e6,1) for r in rows], 'MB fetched per launch (FETCH_SIZE KiB x2)')
PY
done
