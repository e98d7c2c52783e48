import subprocess, shutil, os, sys
BASE='/tmp/base_r03'
os.chdir(BASE)
def sh(c): subprocess.run(c, shell=True, check=True)
def read(p): return open(p).read()
def write(p,s): open(p,'w').write(s)
AH='magicpig_amd/csrc/attn_head.h'; LSH='magicpig_amd/csrc/lsh.hip'
orig_ah=read(AH); orig_lsh=read(LSH)

def p_fence(ah, lsh):
    old='''                vreg[u] = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(row + D));
            }
        }'''
    new='''                vreg[u] = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(row + D));
            }
            __builtin_amdgcn_sched_barrier(0);
        }'''
    assert old in ah; return ah.replace(old,new), lsh

def p_off32(ah, lsh):
    old='''    const uint16_t* kvc = kv_g + c * 8;
'''
    new='''    const char* kvb = reinterpret_cast<const char*>(kv_g);
    const uint32_t coff = (uint32_t)c * 16u;
'''
    assert old in ah; ah=ah.replace(old,new)
    old='''        if (SLICE < AH_SLICE) {
#pragma unroll
            for (int u = 0; u < UPS; ++u)
                kreg[u] = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(kvc + (int64_t)idc[u] * 2 * D));
#pragma unroll
            for (int u = 0; u < UPS; ++u)
                vreg[u] = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(kvc + (int64_t)idc[u] * 2 * D + D));
        } else {
#pragma unroll
            for (int u = 0; u < UPS; ++u) {
                const uint16_t* row = kvc + (int64_t)idc[u] * 2 * D;
                kreg[u] = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(row));
                vreg[u] = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(row + D));
            }'''
    new='''        uint32_t ro[UPS];
#pragma unroll
        for (int u = 0; u < UPS; ++u) ro[u] = (uint32_t)idc[u] * (uint32_t)(4 * D) + coff;
        if (SLICE < AH_SLICE) {
#pragma unroll
            for (int u = 0; u < UPS; ++u)
                kreg[u] = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(kvb + ro[u]));
#pragma unroll
            for (int u = 0; u < UPS; ++u)
                vreg[u] = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(kvb + ro[u] + 2 * D));
        } else {
#pragma unroll
            for (int u = 0; u < UPS; ++u) {
                kreg[u] = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(kvb + ro[u]));
                vreg[u] = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(kvb + ro[u] + 2 * D));
            }'''
    assert old in ah; return ah.replace(old,new), lsh

def p_knlds(ah, lsh):
    old='''        float kn_my = 1.f;
        if (!DENSE) {
            if (kn_lds != nullptr) kn_my = bf16_bits_to_f32(kn_lds[id_my - kn_t0]);
            else kn_my = kn_g[id_my];
        }
'''
    new='''        float kn_my = 1.f;
        if (!DENSE && kn_lds == nullptr) kn_my = kn_g[id_my];
'''
    assert old in ah; ah=ah.replace(old,new)
    old='''        if (!DENSE && k == wave) MP_STAMP(stamp, 34);
'''
    new='''        if (!DENSE && kn_lds != nullptr) kn_my = bf16_bits_to_f32(kn_lds[id_my - kn_t0]);
        if (!DENSE && k == wave) MP_STAMP(stamp, 34);
'''
    assert old in ah; return ah.replace(old,new), lsh

def p_state(ah, lsh):
    old='''    if (AD > 0 && aa.idbits_dev != nullptr) idbits = *aa.idbits_dev;
    const uint32_t idmask = idbits ? ((1u << idbits) - 1u) : 0xffffffffu;
'''
    new='''    uint32_t idmask = idbits ? ((1u << idbits) - 1u) : 0xffffffffu;
'''
    assert old in lsh; lsh=lsh.replace(old,new)
    old='''    if (AD > 0 && idbits != 0 && aa.pay_bad != nullptr) pay = aa.pay_bad[g] == 0 && aa.att_ver[g] == aa.kn_ver[g];
'''
    assert old in lsh; lsh=lsh.replace(old,'')
    old='''    for (int l = tid; l < Lpad; l += RT_THREADS) s_len[l] = 0;
    if (HASH == 2) {'''
    new='''    for (int l = tid; l < Lpad; l += RT_THREADS) s_len[l] = 0;
    if (AD > 0 && aa.idbits_dev != nullptr) {
        const int ib = *aa.idbits_dev;
        int bad = 1;
        unsigned int av = 0u, kv = 1u;
        if (aa.pay_bad != nullptr) {
            bad = aa.pay_bad[g];
            av = aa.att_ver[g];
            kv = aa.kn_ver[g];
        }
        idbits = ib;
        pay = (ib != 0) & (bad == 0) & (av == kv);
        idmask = idbits ? ((1u << idbits) - 1u) : 0xffffffffu;
    }
    if (HASH == 2) {'''
    assert old in lsh; return ah, lsh.replace(old,new)

def p_nostore(ah, lsh):   # fused path: the ascending list goes to HBM only when it spills the LDS stage
    old='''            out[off] = base + p;
            if (AD > 0 && off < aa.cap) s_ids[off] = base + p;'''
    new='''            if (AD == 0 || spill) out[off] = base + p;
            if (AD > 0 && off < aa.cap) s_ids[off] = base + p;'''
    assert old in lsh; return ah, lsh.replace(old,new)

variants={'vfence':[p_fence],'voff32':[p_off32],'vknlds':[p_knlds],'vstate':[p_state],'vnostore':[p_nostore],
          'vall5':[p_fence,p_off32,p_knlds,p_state,p_nostore]}
for name,ps in variants.items():
    ah,lsh=orig_ah,orig_lsh
    for f in ps: ah,lsh=f(ah,lsh)
    write(AH,ah); write(LSH,lsh)
    sh(f'python scripts/build_variant.py {name} > /dev/null')
    os.makedirs(f'/root/repo/magicpig_amd/lib/variants/{name}', exist_ok=True)
    shutil.copy(f'magicpig_amd/lib/variants/{name}/libmagicpig_hip.so', f'/root/repo/magicpig_amd/lib/variants/{name}/')
    sh(f'python /tmp/kres.py {BASE}/magicpig_amd/lib/variants/{name} | grep "decode_kernelILi16ELi128ELb0ELi[13]" | sed "s/^/{name} /"')
write(AH,orig_ah); write(LSH,orig_lsh)
