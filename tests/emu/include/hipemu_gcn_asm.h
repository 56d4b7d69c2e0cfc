// TEST INFRASTRUCTURE -- interprets the handful of gfx950 VALU instructions the library's inline assembly uses
// (galah_amd/csrc/murmur21_asm.h), so that the emulated build executes the SAME instruction text the GPU does.
// tests/emu/gen_sources.py turns each `asm volatile(TEXT : outputs : inputs : clobbers)` into a call of run() with the
// operand bindings it reads off the statement.
#pragma once
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <unordered_map>
#include <vector>

#include <hipemu_wavesan.h>

namespace hipemu_gcn {

struct Bind {   // a named operand: %[name]
    const char *name;
    uint64_t value;
    int bits;   // 32 or 64 (an SGPR / VGPR pair)
};

enum Op { ALIGNBIT, MAD_U64_U32, ADD_U32, LSHL_ADD_U64, XOR_B32, LSHRREV_B32, ADD3_U32, AND_B32, OR_B32, LSHLREV_B32, MOV_B32, SUB_U32, MUL_LO_U32, MUL_HI_U32 };
enum Kind { VREG, VREG64, NAMED, IMM, VCC };
struct Arg { Kind kind; uint32_t reg; int named; uint64_t imm; };
struct Ins { Op op; Arg a[5]; int n; };
struct Program { std::vector<Ins> ins; std::vector<std::string> names; };

inline Program parse(const char *text) {
    static const struct { const char *m; Op op; } table[] = {
        {"v_alignbit_b32", ALIGNBIT}, {"v_mad_u64_u32", MAD_U64_U32}, {"v_add_u32", ADD_U32}, {"v_lshl_add_u64", LSHL_ADD_U64},
        {"v_xor_b32", XOR_B32}, {"v_lshrrev_b32", LSHRREV_B32}, {"v_add3_u32", ADD3_U32}, {"v_and_b32", AND_B32}, {"v_or_b32", OR_B32},
        {"v_lshlrev_b32", LSHLREV_B32}, {"v_mov_b32", MOV_B32}, {"v_sub_u32", SUB_U32}, {"v_mul_lo_u32", MUL_LO_U32}, {"v_mul_hi_u32", MUL_HI_U32}};
    Program p;
    std::string s(text);
    size_t pos = 0;
    while (pos < s.size()) {
        size_t eol = s.find('\n', pos);
        if (eol == std::string::npos) eol = s.size();
        std::string line = s.substr(pos, eol - pos);
        pos = eol + 1;
        size_t c = line.find("/*");
        if (c != std::string::npos) line = line.substr(0, c);
        size_t b = line.find_first_not_of(" \t");
        if (b == std::string::npos) continue;
        size_t e = line.find_first_of(" \t", b);
        std::string mn = line.substr(b, e == std::string::npos ? std::string::npos : e - b);
        Ins ins{};
        bool found = false;
        for (auto &t : table)
            if (mn == t.m) { ins.op = t.op; found = true; }
        if (!found) { fprintf(stderr, "[hipemu_gcn] unknown instruction '%s'\n", mn.c_str()); abort(); }
        std::string rest = e == std::string::npos ? "" : line.substr(e);
        // split operands at commas outside brackets
        std::vector<std::string> ops;
        std::string cur;
        int depth = 0;
        for (char ch : rest) {
            if (ch == '[') depth++;
            if (ch == ']') depth--;
            if (ch == ',' && depth == 0) { ops.push_back(cur); cur.clear(); }
            else if (ch != ' ' && ch != '\t') cur += ch;
        }
        if (!cur.empty()) ops.push_back(cur);
        ins.n = (int)ops.size();
        if (ins.n > 5) abort();
        for (int i = 0; i < ins.n; i++) {
            const std::string &o = ops[i];
            Arg a{};
            if (o == "vcc") a.kind = VCC;
            else if (o.rfind("%[", 0) == 0) {
                std::string nm = o.substr(2, o.size() - 3);
                a.kind = NAMED;
                a.named = -1;
                for (size_t k = 0; k < p.names.size(); k++)
                    if (p.names[k] == nm) a.named = (int)k;
                if (a.named < 0) { a.named = (int)p.names.size(); p.names.push_back(nm); }
            } else if (o.rfind("v[", 0) == 0) { a.kind = VREG64; a.reg = (uint32_t)atoi(o.c_str() + 2); }
            else if (o[0] == 'v') { a.kind = VREG; a.reg = (uint32_t)atoi(o.c_str() + 1); }
            else { a.kind = IMM; a.imm = strtoull(o.c_str(), nullptr, 0); }
            ins.a[i] = a;
        }
        p.ins.push_back(ins);
    }
    return p;
}

struct Machine {
    uint32_t v[256];
    uint64_t named[32];
    uint64_t vcc;
    const Program *prog;
    uint64_t out(const char *name) const {   // a named output operand after run()
        for (size_t k = 0; k < prog->names.size(); k++)
            if (prog->names[k] == name) return named[k];
        abort();
    }
    uint64_t reg64(uint32_t r) const { return (uint64_t)v[r] | ((uint64_t)v[r + 1] << 32); }
    uint64_t rd(const Arg &a, bool wide) const {
        switch (a.kind) {
        case VREG: return v[a.reg];
        case VREG64: return (uint64_t)v[a.reg] | ((uint64_t)v[a.reg + 1] << 32);
        case NAMED: return wide ? named[a.named] : (uint32_t)named[a.named];
        case IMM: return a.imm;
        default: return vcc;
        }
    }
    void wr(const Arg &a, uint64_t x, bool wide) {
        switch (a.kind) {
        case VREG: v[a.reg] = (uint32_t)x; break;
        case VREG64: v[a.reg] = (uint32_t)x; v[a.reg + 1] = (uint32_t)(x >> 32); break;
        case NAMED: named[a.named] = wide ? x : (uint32_t)x; break;
        default: abort();
        }
    }
};

// runs `text` with the named inputs bound; leaves the machine for the caller to read named outputs / fixed registers from
inline void run(const char *text, const Bind *binds, int n_binds, Machine &m) {
    hipemu::WaveSanSuppress not_the_kernels_memory;   // (the parsed program, its cache: the interpreter's own; the instructions touch registers only)
    static std::mutex mu;
    static std::unordered_map<const char *, Program *> cache;
    static thread_local const char *last_text[2] = {nullptr, nullptr};
    static thread_local Program *last_prog[2] = {nullptr, nullptr};
    Program *p;
    if (text == last_text[0]) p = last_prog[0];
    else if (text == last_text[1]) p = last_prog[1];
    else {
        std::lock_guard<std::mutex> lk(mu);
        auto it = cache.find(text);
        if (it == cache.end()) it = cache.emplace(text, new Program(parse(text))).first;
        p = it->second;
        last_text[1] = last_text[0]; last_prog[1] = last_prog[0];
        last_text[0] = text; last_prog[0] = p;
    }
    m.prog = p;
    for (size_t k = 0; k < p->names.size(); k++) {
        m.named[k] = 0;
        for (int i = 0; i < n_binds; i++)
            if (p->names[k] == binds[i].name) m.named[k] = binds[i].value;
    }
    for (const Ins &i : p->ins) {
        const Arg *a = i.a;
        switch (i.op) {
        case ALIGNBIT: m.wr(a[0], (uint32_t)((((uint64_t)(uint32_t)m.rd(a[1], false) << 32) | (uint32_t)m.rd(a[2], false)) >> (m.rd(a[3], false) & 31)), false); break;
        case MAD_U64_U32: {   // D[2], vcc, S0, S1, S2[2]
            const unsigned __int128 r = (unsigned __int128)((uint64_t)(uint32_t)m.rd(a[2], false) * (uint32_t)m.rd(a[3], false)) + m.rd(a[4], true);
            m.wr(a[0], (uint64_t)r, true);
            m.vcc = (uint64_t)(r >> 64) ? ~0ull : 0;
            break;
        }
        case ADD_U32: m.wr(a[0], (uint32_t)m.rd(a[1], false) + (uint32_t)m.rd(a[2], false), false); break;
        case SUB_U32: m.wr(a[0], (uint32_t)m.rd(a[1], false) - (uint32_t)m.rd(a[2], false), false); break;
        case ADD3_U32: m.wr(a[0], (uint32_t)m.rd(a[1], false) + (uint32_t)m.rd(a[2], false) + (uint32_t)m.rd(a[3], false), false); break;
        case LSHL_ADD_U64: m.wr(a[0], (m.rd(a[1], true) << (m.rd(a[2], false) & 7)) + m.rd(a[3], true), true); break;   // shift 0..4 in the ISA
        case XOR_B32: m.wr(a[0], (uint32_t)m.rd(a[1], false) ^ (uint32_t)m.rd(a[2], false), false); break;
        case AND_B32: m.wr(a[0], (uint32_t)m.rd(a[1], false) & (uint32_t)m.rd(a[2], false), false); break;
        case OR_B32: m.wr(a[0], (uint32_t)m.rd(a[1], false) | (uint32_t)m.rd(a[2], false), false); break;
        case LSHRREV_B32: m.wr(a[0], (uint32_t)m.rd(a[2], false) >> (m.rd(a[1], false) & 31), false); break;
        case LSHLREV_B32: m.wr(a[0], (uint32_t)m.rd(a[2], false) << (m.rd(a[1], false) & 31), false); break;
        case MOV_B32: m.wr(a[0], (uint32_t)m.rd(a[1], false), false); break;
        case MUL_LO_U32: m.wr(a[0], (uint32_t)m.rd(a[1], false) * (uint32_t)m.rd(a[2], false), false); break;
        case MUL_HI_U32: m.wr(a[0], (uint32_t)(((uint64_t)(uint32_t)m.rd(a[1], false) * (uint32_t)m.rd(a[2], false)) >> 32), false); break;
        }
    }
}
}  // namespace hipemu_gcn
