"""Python model of what the HIP kernels compute FROM THE INTEGER TABLES the host builder emits
(genz-tokenize_amd/csrc/gz_tables.cpp).  Test infrastructure: it lets the CPU-only suite check the
table builder and the kernels' closed-form pad / pair formulas against the oracle without a GPU.
It mirrors gz_kernels.hip step by step (same perfect-hash probe, same symbol encoding, same formulas)."""
import numpy as np

WS = frozenset([0x09, 0x0A, 0x0B, 0x0C, 0x0D, 0x1C, 0x1D, 0x1E, 0x1F, 0x20, 0x85, 0xA0, 0x1680, 0x2028, 0x2029,
                0x202F, 0x205F, 0x3000] + list(range(0x2000, 0x200B)))
UNKNOWN = 0x80000000
NONE = -1
M32 = 0xFFFFFFFF


def cp_hash(cp):
    h = (cp * 0x9E3779B1) & M32
    return h ^ (h >> 16)


class TableSim:
    def __init__(self, H):
        self.merges = H.array(1)
        self.sym_ids = H.array(2)
        self.bmp = H.array(3)
        self.astral = H.array(4)
        self.pad, self.bos, self.eos, _, self.unk = [int(x) for x in H.array(5)]
        # the perfectly hashed form (gz_common.h): entries, displacement array, description, hot set
        self.pair8 = H.array(6)
        self.disp = H.array(7)
        self.nbuckets, self.bshift, self.sshift, self.slots, self.k1, self.k2, self.n_overflow = [int(x) for x in H.array(8)]
        self.hot = H.array(9)

    PH_MUL, PH_OVERFLOW, HOT_SHIFT = 0x2C1B3C6D, 0xFFFF, 20

    def probe8(self, a, b, use_hot=True):
        """gz_miss2_kernel's lookup: hot set (LDS), else displacement -> ONE slot; overflow buckets probe on.  Returns
        (rank, alias flag) or None."""
        if (a | b) & 0xFFF00000:
            return None
        klo, khi = (a | (b << 20)) & M32, b >> 12
        ha = (a * self.k1 + b * self.k2) & M32
        hb = (a * 0xC2B2AE36 + b * 0x27D4EB2F) & M32

        def match(e):
            return int(e[0]) == klo and (int(e[1]) & 0xFF) == khi
        if use_hot:
            e = self.hot[ha >> self.HOT_SHIFT]
            if match(e):
                return int(e[1]) >> 9, (int(e[1]) >> 8) & 1
        d = int(self.disp[ha >> self.bshift])
        slot = (((hb ^ d) * self.PH_MUL) & M32) >> self.sshift
        e = self.pair8[slot]
        if match(e):
            return int(e[1]) >> 9, (int(e[1]) >> 8) & 1
        if d != self.PH_OVERFLOW:
            return None
        while int(e[0]) != M32 or int(e[1]) != M32:
            slot = (slot + 1) & (self.slots - 1)
            e = self.pair8[slot]
            if match(e):
                return int(e[1]) >> 9, (int(e[1]) >> 8) & 1
        return None

    def probe(self, a, b):
        """probe_pair of gz_kernels.hip (displacement from memory, no hot set): the rank, or None.  The merged symbol the
        kernels take -- the rank, or merges[rank].merged under the alias flag -- must be what the merge list says."""
        r = self.probe8(a, b, use_hot=False)
        if r is None:
            return None
        rank, alias = r
        merged = int(self.merges[rank][2])
        assert (merged != rank) == bool(alias)
        return rank

    def initial(self, cp, last):
        s = M32
        if cp < 0x10000:
            s = int(self.bmp[cp][1 if last else 0])
        elif len(self.astral):
            m = len(self.astral) - 1
            h = cp_hash(cp) & m
            while True:
                e = self.astral[h]
                if int(e[0]) == cp:
                    s = int(e[2 if last else 1]); break
                if int(e[0]) == M32:
                    break
                h = (h + 1) & m
        return (UNKNOWN | cp) if s == M32 else s

    def merge(self, syms):
        while len(syms) > 1:
            best = None
            for a, b in zip(syms, syms[1:]):
                r = self.probe(a, b)
                if r is not None and (best is None or r < best):
                    best = r
            if best is None:
                break
            left, right, merged = (int(x) for x in self.merges[best][:3])
            out, k = [], 0
            while k < len(syms):
                if k + 1 < len(syms) and syms[k] == left and syms[k + 1] == right:
                    out.append(merged); k += 2
                else:
                    out.append(syms[k]); k += 1
            syms = out
        return syms

    def token_id(self, s, final):
        if s & UNKNOWN:
            return self.unk
        return int(self.sym_ids[s][1 if final else 0])

    def words(self, text):
        cps = [ord(c) for c in text]
        i, n, out = 0, len(cps), []
        while i < n:
            if cps[i] in WS:
                i += 1; continue
            j = i
            while j < n and cps[j] not in WS:
                j += 1
            glue = j < n and cps[j] == 0x0A
            out.append((cps[i:j], glue))
            i = j + (1 if glue else 0)
        return out

    def word_symbols(self, cps, glue):
        syms = [self.initial(c, (not glue) and k == len(cps) - 1) for k, c in enumerate(cps)]
        if glue:
            syms.append(self.initial(0x0A, True))
        return self.merge(syms)

    def raw_ids(self, text, pair):
        ids = [self.bos]
        for cps, glue in self.words(text):
            s = self.word_symbols(cps, glue)
            ids += [self.token_id(x, k == len(s) - 1) for k, x in enumerate(s)]
        ids.append(self.eos)
        if pair is not None:
            ids.append(self.eos)
            for cps, glue in self.words(pair):
                s = self.word_symbols(cps, glue)
                ids += [self.token_id(x, k == len(s) - 1) for k, x in enumerate(s)]
            ids.append(self.eos)
        return ids

    # ---- the closed forms of gz_finalize_kernel / gz_pair_kernel ------------------------------------------
    @staticmethod
    def cut_len(n, max_len):
        stop = max_len - 1
        if stop >= 0:
            return min(n, stop)
        return max(0, n + stop)

    def shape_row(self, raw, max_len, padding, truncation):
        pad_mode = max_len is not None and bool(padding)
        n = len(raw)
        if not pad_mode:
            return list(raw)
        if n < max_len:
            return list(raw) + [self.pad] * (max_len - n)
        if truncation:
            k = self.cut_len(n, max_len)
            return list(raw[:k]) + [self.eos]
        return list(raw)

    def pair_rows(self, ids, max_len, padding, truncation):
        R = len(ids)
        eos, bos = self.eos, self.bos
        p1 = next((i for i in range(R) if ids[i] == eos), R)
        seq_len = R
        for i in range(p1 + 2, R):
            if ids[i] == eos and ids[i - 1] != eos:
                seq_len = i + 1; break

        def raw_val(i):
            if i < p1:
                return NONE if ids[i] == bos else 0
            if i == p1:
                return NONE
            return NONE if ids[i] == eos else 1
        nones = [i for i in range(1, seq_len - 1) if raw_val(i) == NONE][:2]
        if len(nones) < 2:
            return None

        def final_val(i):
            if i == seq_len - 1: return 1
            if i == 0: return 0
            if i == nones[0]: return 0
            if i == nones[1]: return 1
            return raw_val(i)
        seq = [final_val(i) for i in range(seq_len)]
        pad_mode = max_len is not None and bool(padding)
        tt = list(seq)
        if pad_mode:
            if seq_len < max_len:
                tt = seq + [self.pad] * (max_len - seq_len)
            elif truncation:
                tt = seq[:self.cut_len(seq_len, max_len)] + [eos]
        return seq, tt

    def call(self, text, pair=None, max_len=None, padding=True, truncation=True):
        ids = self.shape_row(self.raw_ids(text, pair), max_len, padding, truncation)
        res = {"input_ids": ids, "attention_mask": [1 if v != self.pad else 0 for v in ids]}
        if pair is not None:
            pr = self.pair_rows(ids, max_len, padding, truncation)
            if pr is None:
                raise ValueError("None is not in list")
            res["sequence_id"] = [None if v == NONE else v for v in pr[0]]
            res["token_type_ids"] = [None if v == NONE else v for v in pr[1]]
        return res
