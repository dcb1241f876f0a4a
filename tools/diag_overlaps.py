#!/usr/bin/env python3
"""Exploration aid (not a test): the reference hifiasm-0.14's own final overlaps (its <prefix>.ovlp.source.bin dump: ma_hit_t records,
Overlaps.cpp write_ma) and read layout (A lines of p_ctg.gfa) next to the oracle's final overlaps (ORC_DEBUG_HITS=1) for one read set:
    python tools/diag_overlaps.py <region> <hap> <width> <depth>        e.g. 531 1 26000 8.0
Prints the pairs whose coordinates / exact flag differ.  Needs oracle/_ref (built from /root/reference)."""
import sys, os, struct, subprocess, tempfile, re
sys.path.insert(0,'/root/repo')
from focalsv_amd import synth
from tests import oracle_lib as O
region, hap, width, depth = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), float(sys.argv[4])
r=synth.make_region(region, width=width, depth_per_hap=depth)
tmp=tempfile.mkdtemp()
d=synth.write_region_dir(r, tmp)
subprocess.run(["/root/repo/oracle/_ref/hifiasm-0.14","-f0","-o","x.asm","-t","8","PS1_hp%d.fa"%hap],cwd=d,stdout=subprocess.DEVNULL,stderr=subprocess.DEVNULL,check=True)
f=open(os.path.join(d,'x.asm.ovlp.source.bin'),'rb')
n,=struct.unpack('<q',f.read(8))
ref={}
for i in range(n):
    fc,ab,ln=struct.unpack('<BBI',f.read(6))
    for k in range(ln):
        qns,qe,tn,ts,te,el,nli,ml,rev,bl,dl=struct.unpack('<QIIIIBBIIII',f.read(42))
        ref[(qns>>32,tn)]=(qns&0xffffffff,qe,ts,te,rev,el)
print("hifiasm layout:")
for l in open(os.path.join(d,'x.asm.p_ctg.gfa')):
    if l[0]=='A': print("  ",l.split()[1:6])
# mine
import io
rd, wr = os.pipe()
env=dict(os.environ, ORC_DEBUG_HITS="1")
code="import sys; sys.path.insert(0,'/root/repo'); from focalsv_amd import synth; from tests import oracle_lib as O; r=synth.make_region(%d,width=%d,depth_per_hap=%r); c,_=O.assemble(r.reads[%d],O.default_params()); print('CONTIGS',[len(x) for x in c])"%(region,width,depth,hap-1)
p=subprocess.run([sys.executable,"-c",code],env=env,capture_output=True,text=True)
mine={}
for l in p.stderr.splitlines():
    if l.startswith("HIT"):
        q,t,xs,xe,ys,ye,rev,ex=map(int,l.split()[1:])
        mine[(q,t)]=(xs,xe,ys,ye,rev,ex)
print(p.stdout.strip())
nd=0
for k in sorted(set(ref)|set(mine)):
    a,b=ref.get(k),mine.get(k)
    if a!=b:
        nd+=1
        print("DIFF",k,"hifiasm",a,"mine",b)
print("pairs: hifiasm",len(ref),"mine",len(mine),"differing",nd)
