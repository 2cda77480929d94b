"""CPU stand-ins for the GPU index classes, backed by the oracle: just enough of the VectorIndex / BM25SearchIndex surface for the HOST shells of the
product (comet_amd.index.VectorSearch / TextSearch, comet_amd.hybrid.HybridSearch — the reference's Execute() bodies: node lookup, aggregation over
several queries, LimitResults, Autocut, fusion) to run without a GPU. Test infrastructure only; the per-query searches behind search_batch are the
oracle's, so these tests check the shells, not the kernels."""
import numpy as np

import oracle_lib as orc
from comet_amd.index import DocTokenStore, TextSearch, VectorSearch


class OracleVectorIndex:
    def __init__(self, o, dim, kind, default_nprobes=1):
        self.o, self.dim, self.kind_name, self._np = o, int(dim), kind, default_nprobes
        self.rows = {}

    def add(self, v, i=0):
        """Add(NewVectorNode(v)): ids assigned from 1 when not given (the reference's node ids come from a process-wide counter)"""
        i = i or (max(self.rows) + 1 if self.rows else 1)
        v = np.asarray(v, np.float32)
        rc = self.o.add(i, v)
        assert rc == 0, rc
        if self.o.metric == "cosine":
            v = v / np.float32(np.sqrt(np.float32((v * v).sum())))
        self.rows[i] = v
        return i

    def __len__(self): return len(self.rows)
    def default_nprobes(self): return self._np
    def _check_searchable(self): pass
    def _k_cap(self, k, nprobes): return max(1, len(self) if (k <= 0 or k > len(self)) else k)
    def new_search(self): return VectorSearch(self)

    def _lookup_node_vectors(self, node_ids):
        for i in node_ids:
            if i not in self.rows:
                raise KeyError(f"node {i} not found")
        return [self.rows[i].copy() for i in node_ids]

    def search_batch(self, Q, k, threshold=0.0, nprobes=0, ef_search=0, document_ids=(), k_cap=None, mode=0):
        B = len(Q)
        ids = np.zeros((B, k_cap), np.uint32); sc = np.zeros((B, k_cap), np.float32); cnt = np.zeros(B, np.int32)
        for b, q in enumerate(Q):
            if self.kind_name in ("ivf", "ivfpq"):
                n, gi, gs = self.o.search(q, k, nprobes, threshold=threshold, filter_ids=document_ids, cap=k_cap)
            elif self.kind_name == "hnsw":
                n, gi, gs = self.o.search(q, k, ef_search, threshold=threshold, filter_ids=document_ids, cap=k_cap)
            else:
                n, gi, gs = self.o.search(q, k, threshold=threshold, filter_ids=document_ids, cap=k_cap)
            if n < 0:
                raise ValueError(f"oracle search failed: {n}")
            m = min(n, k_cap)
            ids[b, :m], sc[b, :m], cnt[b] = gi[:m], gs[:m], m
        return ids, sc, cnt


class OracleTextIndex(DocTokenStore):
    """token ids per distinct lower-case word (plain ASCII text: the reference's normalize + tokenize reduce to a split on spaces); the docTokens / deletedDocs
    bookkeeping is the product's own (comet_amd.index.DocTokenStore, what BM25SearchIndex mixes in)"""

    def __init__(self):
        self.o, self.vocab, self.n = orc.BM25(), {}, 0
        self._dt_init()

    def tok(self, text):
        return [self.vocab.setdefault(w, len(self.vocab) + 1) for w in text.split(" ") if w]

    def add(self, doc_id, text):
        toks = self.tok(text)
        self.o.add(int(doc_id), toks); self.n += 1
        self._dt_add(doc_id, toks)

    def remove(self, doc_id):
        assert self.o.remove(int(doc_id)) == 0
        self._dt_remove(doc_id)

    def flush(self):
        self.o.flush(); self._dt_flush()

    def num_docs(self): return self.n
    def new_search(self): return TextSearch(self)

    def search_batch(self, queries, k, document_ids=(), k_cap=None):
        B = len(queries)
        ids = np.zeros((B, k_cap), np.uint32); sc = np.zeros((B, k_cap), np.float32); sc64 = np.zeros((B, k_cap), np.float64); cnt = np.zeros(B, np.int32)
        for b, q in enumerate(queries):
            n, gi, s32, s64 = self.o.search(q, k, filter_ids=document_ids)
            m = min(n, k_cap)
            ids[b, :m], sc[b, :m], sc64[b, :m], cnt[b] = gi[:m], s32[:m], s64[:m], m
        return ids, sc, sc64, cnt
