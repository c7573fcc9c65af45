"""Symbolises return addresses of a crash log WITHOUT a core: a library's load base is page aligned, so a frame's low
12 bits are those of its file offset; for a group of frames assumed to lie in one library every page-aligned base is
tried and kept only if EVERY frame is preceded by a call instruction (e8 rel32 or an ff /2 form).  Seven frames pin a
base uniquely.  Used on gpurun_out/st.log of round 1 (profiles/r1_stream_push_sigsegv_symbolised.txt):
    python tools/symbolise_frames.py /opt/rocm/lib/libamdhip64.so /opt/rocm/lib/libhsa-runtime64.so.1 ...
Edit `groups` for another log."""
import subprocess, sys, struct, bisect, glob, os
groups={"A":[0x757267760607,0x757267756c46,0x75726740f21d,0x75726741290b,0x75726741b635,0x75726741b6a6,0x7572674414d1],
        "B":[0x7573ba836266,0x7573ba8275c0],
        "C":[0x7573c56821f4],
        "PC":[0x7573c531f2fb]}
def segments(path):
    out=subprocess.run(["readelf","-lW",path],capture_output=True,text=True).stdout
    segs=[]
    for l in out.splitlines():
        p=l.split()
        if p and p[0]=="LOAD":
            off,va,fsz,flags=int(p[1],16),int(p[2],16),int(p[4],16)," ".join(p[6:-1])
            segs.append((off,va,fsz,"E" in flags))
    return segs
def is_call_before(data,segs,o):
    # file offset of vaddr o
    for off,va,fsz,ex in segs:
        if ex and va<=o<va+fsz:
            f=off+(o-va)
            b=data[f-7:f]
            if b[2]==0xe8: return True                       # call rel32
            for n in (2,3,6,7):                               # ff /2 forms
                if b[7-n]==0xff and (b[7-n+1]>>3)&7==2: return True
            return False
    return False
libs=sys.argv[1:]
for path in libs:
    try:
        segs=segments(path); data=open(path,"rb").read()
    except Exception as e:
        continue
    ex=[(va,va+fsz) for off,va,fsz,e in segs if e]
    if not ex: continue
    lo,hi=min(a for a,_ in ex),max(b for _,b in ex)
    for g,frames in groups.items():
        if g=="PC": continue
        mn,mx=min(frames),max(frames)
        if mx-mn>hi-lo: continue
        cands=[]
        b0=(mn-hi)&~0xfff
        b=b0
        while b<=mn-lo:
            if all(is_call_before(data,segs,f-b) for f in frames): cands.append(b)
            b+=0x1000
        if cands and len(cands)<=3:
            print(os.path.basename(path),"group",g,"bases",[hex(c) for c in cands],"offsets",[hex(f-cands[0]) for f in frames])
        elif cands:
            print(os.path.basename(path),"group",g,len(cands),"candidate bases (ambiguous)")
