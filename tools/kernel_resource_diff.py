"""tools/kernel_resource_diff.py — registers / spills / scratch of every kernel, this tree against a git revision (default: the
end of round 2): which kernels changed their budget?  (cross-compiles, no GPU; found the chain-kernel occupancy regression of round 3)"""
import re, subprocess, os, sys
ROOT='/root/repo'; T='/tmp/lanes_tmp/cmp'
os.makedirs(T, exist_ok=True)
FLAGS=["--offload-arch=gfx950","-O3","-std=c++17","-fPIC","-ffp-contract=off","-fgpu-flush-denormals-to-zero","--cuda-device-only","-S"]
def res(path, inc):
    out=T+'/'+os.path.basename(path)+'.s'   # (never next to the sources: make has a built-in rule  %: %.s)
    r=subprocess.run(['/opt/rocm/bin/hipcc']+FLAGS+['-I'+inc,'-I'+ROOT+'/include',path,'-o',out],stderr=subprocess.PIPE)
    if r.returncode: return None
    text=open(out).read(); d={}
    for m in re.finditer(r"\.name:\s+(\S+)(.*?)\.wavefront_size", text, re.S):
        b=m.group(2); g=lambda k:int(re.search(r"\.%s:\s+(\d+)"%k,b).group(1))
        d[m.group(1)]=(g('vgpr_count'),g('vgpr_spill_count'),g('private_segment_fixed_size'))
    return d
old=sys.argv[1] if len(sys.argv) > 1 else '236a9b1'
inc_old=T+'/inc_old'; os.makedirs(inc_old,exist_ok=True)
files=subprocess.check_output(['git','-C',ROOT,'ls-tree','--name-only',old,'web-audio-api-rs_amd/csrc/']).decode().split()
for f in files:
    if f.endswith('.hpp'):
        open(inc_old+'/'+os.path.basename(f),'wb').write(subprocess.check_output(['git','-C',ROOT,'show',old+':'+f]))
for f in files:
    if not f.endswith('.hip'): continue
    b=os.path.basename(f)
    po=T+'/old_'+b; open(po,'wb').write(subprocess.check_output(['git','-C',ROOT,'show',old+':'+f]))
    ro=res(po,inc_old); rn=res(ROOT+'/'+f, ROOT+'/web-audio-api-rs_amd/csrc')
    if ro is None or rn is None: print(b,'compile problem'); continue
    for k in sorted(set(ro)&set(rn)):
        if ro[k]!=rn[k]: print(b, k[:90], 'old',ro[k],'new',rn[k])
