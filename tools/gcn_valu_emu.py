# Minimal emulator of the gfx950 VALU instructions hipcc emits for the field arithmetic (v_mad_u64_u32, 64-bit shifts, ...).
# Used to localise a compiler miscompile (see the FE_HIDE24 note in ecloop_amd/csrc/fe256.h): run it over the kernel body of a
# `hipcc -S --cuda-device-only` listing of a kernel that loads 9 limbs, computes fe_sqr(fe_sqr(x)) and stores 9 limbs.
import re, sys, random
M32=0xffffffff; M64=(1<<64)-1
P=2**256-2**32-977
def run(lines, limbs):
    v=[0]*256; s=[0]*104
    out={}
    def rd(op, wide=False):
        op=op.strip()
        m=re.fullmatch(r'v\[(\d+):(\d+)\]',op)
        if m: a=int(m.group(1)); return v[a]|(v[a+1]<<32)
        m=re.fullmatch(r's\[(\d+):(\d+)\]',op)
        if m: a=int(m.group(1)); return s[a]|(s[a+1]<<32)
        if op[0]=='v': return v[int(op[1:])]
        if op[0]=='s' and op[1:].isdigit(): return s[int(op[1:])]
        return int(op,0)&M64
    def wr(op,val,wide=False):
        op=op.strip()
        m=re.fullmatch(r'v\[(\d+):(\d+)\]',op)
        if m:
            a=int(m.group(1)); n=int(m.group(2))-a+1
            for i in range(n): v[a+i]=(val>>(32*i))&M32
            return
        if op[0]=='v': v[int(op[1:])]=val&M32; return
        if op[0]=='s': s[int(op[1:])]=val&M32; return
        raise Exception(op)
    for ln in lines:
        ln=ln.split(';')[0].strip()
        if not ln or ln.endswith(':'): continue
        parts=ln.split(None,1); ins=parts[0]; ops=[o.strip() for o in parts[1].split(',')] if len(parts)>1 else []
        if ins in('s_load_dwordx4','s_waitcnt','s_endpgm','s_nop'): continue
        if ins=='v_mul_u32_u24_e32': wr(ops[0],0); continue   # thread 0
        if ins=='global_load_dwordx4':
            off=int(ops[-1].split('offset:')[1]) if 'offset' in ops[-1] else 0
            base=off//4; wr(ops[0], sum(limbs[base+i]<<(32*i) for i in range(4))); continue
        if ins=='global_load_dword':
            off=int(ops[-1].split('offset:')[1]) if 'offset' in ops[-1] else 0
            wr(ops[0], limbs[off//4]); continue
        if ins.startswith('global_store'):
            last=ops[-1]; off=int(last.split('offset:')[1]) if 'offset' in last else 0
            n={'global_store_dwordx4':4,'global_store_dwordx3':3,'global_store_dwordx2':2,'global_store_dword':1}[ins]
            val=rd(ops[1]) if n>1 else rd(ops[1])
            m=re.fullmatch(r'v\[(\d+):(\d+)\]',ops[1]); a=int(m.group(1))
            for i in range(n): out[off//4+i]=v[a+i]
            continue
        if ins in('s_movk_i32','s_mov_b32'): wr(ops[0], int(ops[1],0)&M32); continue
        if ins=='v_mov_b32_e32': wr(ops[0], rd(ops[1])&M32); continue
        if ins=='v_mov_b64_e32': wr(ops[0], rd(ops[1])&M64); continue
        if ins=='v_lshlrev_b32_e32': wr(ops[0], (rd(ops[2])<<(rd(ops[1])&31))&M32); continue
        if ins=='v_lshrrev_b32_e32': wr(ops[0], (rd(ops[2])&M32)>>(rd(ops[1])&31)); continue
        if ins=='v_and_b32_e32': wr(ops[0], rd(ops[1])&rd(ops[2])&M32); continue
        if ins=='v_add_u32_e32': wr(ops[0], (rd(ops[1])+rd(ops[2]))&M32); continue
        if ins=='v_mad_u64_u32':
            a=rd(ops[2])&M32; b=rd(ops[3])&M32; c=rd(ops[4])&M64
            wr(ops[0], (a*b+c)&M64); continue
        if ins=='v_mad_u32_u24': wr(ops[0], ((rd(ops[1])&0xffffff)*(rd(ops[2])&0xffffff)+rd(ops[3]))&M32); continue
        if ins=='v_lshl_add_u64':
            a=rd(ops[1])&M64; sh=rd(ops[2])&7; c=rd(ops[3])&M64   # shift amount: only 0..4 valid in ISA
            wr(ops[0], ((a<<sh)+c)&M64); continue
        if ins=='v_lshrrev_b64': wr(ops[0], (rd(ops[2])&M64)>>(rd(ops[1])&63)); continue
        if ins=='v_lshlrev_b64': wr(ops[0], ((rd(ops[2])&M64)<<(rd(ops[1])&63))&M64); continue
        if ins=='v_alignbit_b32':
            hi=rd(ops[1])&M32; lo=rd(ops[2])&M32; sh=rd(ops[3])&31
            wr(ops[0], (((hi<<32)|lo)>>sh)&M32); continue
        raise Exception("unhandled "+ln)
    return [out[i] for i in range(9)]
def to_limbs(x): return [(x>>(29*i))&0x1fffffff for i in range(9)]
def val(l): return sum(l[i]<<(29*i) for i in range(9))%P
for name in sys.argv[1:]:
    lines=open(name).read().split('\n')
    rnd=random.Random(1); bad=0
    for x in [1,2,P-1,3]+[rnd.randrange(P) for _ in range(20)]:
        r=run(lines,to_limbs(x))
        if val(r)!=pow(x,4,P): bad+=1
    print(name,"emulated mismatches:",bad)
