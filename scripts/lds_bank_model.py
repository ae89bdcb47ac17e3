#!/usr/bin/env python3
"""Bank-conflict model of the pair FFT's LDS traffic (N = 8192 float2 points, 512 threads, radices 16 16 8 4: the headline
shape's k_fwd_pair_ps), per MI355X_MICROARCH.md's LDS table: ds_write_b64 = 4 groups of 16 contiguous lanes on 32 banks
of 4 bytes, ds_read_b64 = 2 groups of 32 lanes on 64 banks.  Prints LDS-array cycles per transform for the layout the
kernels use (one pad element per 32: csrc/fft_lds.h phys()) and for the alternatives.

Result (profiles/r03_lds_model.txt): the shipped layout spends 3648 cycles per transform where a conflict-free one
would spend 3072 (+19 %: the 2-way conflict of the first pass's stride-16 writes, 512 cycles, and 64 in the split step's
mirrored reads).  The SQ counters of round 2 say 769 conflict cycles per transform, so the model is about right.  No
single layout is free of both: padding per 16 moves the conflict to the reads (+35 %), the XOR swizzle trades the
first pass for the split step (+10 %) at one more address instruction per access.  An LDS array that is 25 % busy
(3648 x 64 transforms per CU and launch = 0.10 of the kernel's 0.40 ms) is not what these kernels wait for -- round 1
and 2 showed them bound by the chain of dependent phases at two workgroups per CU -- so the 0.016 ms the conflicts cost
at most were left alone."""
import sys
LOG2N=13; M=1<<LOG2N; NT=512; P=M//NT
RAD=[16,16,8,4]; NP=4
def pprod(s):
    p=1
    for j in range(s): p*=RAD[j]
    return p
def widx(S,tid,b,q):
    R=RAD[S]; p=pprod(S); i=tid+b*NT; k=i&(p-1); return (i-k)*R+k+q*p
def ridx(S,tid,b,r):
    return (tid+b*NT)+r*(M//RAD[S+1])
def cycles(addrs_bytes, kind):
    # addrs_bytes: list of 64 byte addresses (8-byte accesses)
    tot=0
    if kind=='w':
        groups=[range(g*16,(g+1)*16) for g in range(4)]; nb=32
    else:
        groups=[range(0,32),range(32,64)]; nb=64
    for g in groups:
        banks={}
        for l in g:
            a=addrs_bytes[l]
            for d in (a//4, a//4+1):
                banks.setdefault(d%nb,set()).add(d)
        tot+=max(len(v) for v in banks.values())
    return tot
def run(phys, name):
    total=0; ideal=0
    out=[]
    for S in range(NP-1):
        R=RAD[S]; Rn=RAD[S+1]
        cw=0; iw=0
        for b in range(P//R):
            for q in range(R):
                for wv in range(NT//64):
                    a=[phys(widx(S,wv*64+l,b,q))*8 for l in range(64)]
                    cw+=cycles(a,'w'); iw+=4
        cr=0; ir=0
        for b in range(P//Rn):
            for r in range(Rn):
                for wv in range(NT//64):
                    a=[phys(ridx(S,wv*64+l,b,r))*8 for l in range(64)]
                    cr+=cycles(a,'r'); ir+=2
        out.append((S,cw,iw,cr,ir)); total+=cw+cr; ideal+=iw+ir
    # forward split step: write Z natural order at phys(out_index) ; reads zk (2 adjacent -> model as two b64), zn
    def out_index(tid,e):
        R=RAD[NP-1]; inv=lambda p:p  # radix-4 pos identity
        return (tid+(e//R)*NT)+inv(e%R)*(M//R)
    cw=0; iw=0
    for e in range(P):
        for wv in range(NT//64):
            a=[phys(out_index(wv*64+l,e))*8 for l in range(64)]
            cw+=cycles(a,'w'); iw+=4
    cr=0; ir=0
    for j in range(P//4):
        for wv in range(NT//64):
            for off,sign in ((0,1),(1,1)):
                a=[phys(2*(wv*64+l)+2*NT*j+off)*8 for l in range(64)]
                cr+=cycles(a,'r'); ir+=2
            for off in (0,1):
                a=[phys((M-(2*(wv*64+l)+2*NT*j+off))&(M-1) if True else 0)*8 for l in range(64)]
                cr+=cycles(a,'r'); ir+=2
    out.append(('split',cw,iw,cr,ir)); total+=cw+cr; ideal+=iw+ir
    print(name, "total LDS cycles per transform %d (conflict-free %d, +%.0f%%)"%(total,ideal,100*(total-ideal)/ideal))
    for o in out: print("   pass",o[0],"write %d/%d  read %d/%d"%(o[1],o[2],o[3],o[4]))
run(lambda i:i+(i>>5),"pad 1 per 32:")
run(lambda i:i,"no padding:")
run(lambda i:i^((i>>4)&15),"xor swizzle:")
run(lambda i:i+(i>>4),"pad 1 per 16:")
run(lambda i:i+(i>>5)+(i>>9),"pad 1/32 + 1/512:")
run(lambda i:i+(i>>4)+(i>>8),"pad 1/16 + 1/256:")
