package cometgpu

/*
#include "comet_gpu.h"
*/
import "C"

import (
	"encoding/binary"
	"fmt"
	"io"
	"math"
	"sort"
	"strings"
	"sync"

	"github.com/RoaringBitmap/roaring"
	"github.com/clipperhouse/uax29/v2/words"
	comet "github.com/wizenheimer/comet"
	"golang.org/x/text/unicode/norm"
)

// BM25SearchIndex implements comet.TextIndex over the GPU BM25 kernels. Tokenisation (NFKC + lower-casing + uax29 word
// segmentation, bm25_index.go:154-166) stays here in Go with the reference's own dependencies; documents and queries
// cross the C ABI as token ids. The token lists are kept on the host like the reference's docTokens (bm25_index.go:107):
// they serve WithNode queries and the on-disk format.
type BM25SearchIndex struct {
	mu        sync.RWMutex
	ctx       *Context
	h         *C.comet_text_index
	vocab     map[string]uint32
	docTokens map[uint32][]string
	deleted   map[uint32]struct{}
}

var _ comet.TextIndex = (*BM25SearchIndex)(nil)

// NewBM25SearchIndex mirrors comet.NewBM25SearchIndex() (bm25_index.go:143).
func NewBM25SearchIndex(ctx *Context) (*BM25SearchIndex, error) {
	ix := &BM25SearchIndex{ctx: ctx, vocab: map[string]uint32{}, docTokens: map[uint32][]string{}, deleted: map[uint32]struct{}{}}
	if rc := C.comet_bm25_create(ctx.h, &ix.h); rc != C.COMET_OK {
		return nil, lastError(rc)
	}
	return ix, nil
}

func (ix *BM25SearchIndex) Close() {
	ix.mu.Lock()
	defer ix.mu.Unlock()
	if ix.h != nil {
		C.comet_bm25_destroy(ix.h)
		ix.h = nil
	}
}

func normalize(s string) string { return strings.ToLower(norm.NFKC.String(s)) } // bm25_index.go:154

func tokenize(s string) []string { // bm25_index.go:159
	toks := words.FromString(s)
	var tokens []string
	for toks.Next() {
		tokens = append(tokens, toks.Value())
	}
	return tokens
}

// tokenIDs maps tokens to dense ids; unknown tokens get fresh ids when `grow` is set, else the id ^uint32(0) (no postings).
func (ix *BM25SearchIndex) tokenIDs(tokens []string, grow bool) []uint32 {
	out := make([]uint32, len(tokens))
	for i, t := range tokens {
		id, ok := ix.vocab[t]
		if !ok {
			if grow {
				id = uint32(len(ix.vocab))
				ix.vocab[t] = id
			} else {
				id = math.MaxUint32
			}
		}
		out[i] = id
	}
	return out
}

// Add indexes a document (bm25_index.go:168-201); re-adding an id replaces the old document.
func (ix *BM25SearchIndex) Add(id uint32, text string) error {
	tokens := tokenize(normalize(text))
	ix.mu.Lock()
	defer ix.mu.Unlock()
	return ix.addTokens(id, tokens)
}

func (ix *BM25SearchIndex) addTokens(id uint32, tokens []string) error {
	tids := ix.tokenIDs(tokens, true)
	var p *C.uint32_t
	if len(tids) > 0 {
		p = (*C.uint32_t)(&tids[0])
	}
	if rc := C.comet_bm25_add(ix.h, C.uint32_t(id), p, C.int32_t(len(tids))); rc != C.COMET_OK {
		return lastError(rc)
	}
	ix.docTokens[id] = tokens
	delete(ix.deleted, id)
	return nil
}

// Remove soft-deletes (bm25_index.go:203).
func (ix *BM25SearchIndex) Remove(id uint32) error {
	ix.mu.Lock()
	defer ix.mu.Unlock()
	if rc := C.comet_bm25_remove(ix.h, C.uint32_t(id)); rc != C.COMET_OK {
		return lastError(rc)
	}
	ix.deleted[id] = struct{}{}
	return nil
}

// Flush hard-deletes (bm25_index.go:374).
func (ix *BM25SearchIndex) Flush() error {
	ix.mu.Lock()
	defer ix.mu.Unlock()
	if rc := C.comet_bm25_flush(ix.h); rc != C.COMET_OK {
		return lastError(rc)
	}
	for id := range ix.deleted {
		delete(ix.docTokens, id)
	}
	ix.deleted = map[uint32]struct{}{}
	return nil
}

func (ix *BM25SearchIndex) NewSearch() comet.TextSearch { return &textSearch{index: ix, k: 10, cutoff: -1} }

// ---- TextSearch ------------------------------------------------------------------------------------------------------
type textSearch struct {
	index       *BM25SearchIndex
	queries     []string
	nodeIDs     []uint32
	k           int
	aggregation comet.ScoreAggregationKind
	cutoff      int
	documentIDs []uint32
}

var _ comet.TextSearch = (*textSearch)(nil)

func (s *textSearch) WithQuery(queries ...string) comet.TextSearch { s.queries = queries; return s }
func (s *textSearch) WithNode(nodeIDs ...uint32) comet.TextSearch  { s.nodeIDs = nodeIDs; return s }
func (s *textSearch) WithK(k int) comet.TextSearch                 { s.k = k; return s }
func (s *textSearch) WithScoreAggregation(kind comet.ScoreAggregationKind) comet.TextSearch {
	s.aggregation = kind
	return s
}
func (s *textSearch) WithCutoff(cutoff int) comet.TextSearch             { s.cutoff = cutoff; return s }
func (s *textSearch) WithDocumentIDs(docIDs ...uint32) comet.TextSearch { s.documentIDs = docIDs; return s }

// Execute mirrors bm25TextSearch.Execute (bm25_index_search.go:152-190): all queries in ONE device call.
func (s *textSearch) Execute() ([]comet.TextResult, error) {
	if len(s.queries) == 0 && len(s.nodeIDs) == 0 {
		return nil, fmt.Errorf("must specify either queries or node IDs")
	}
	aggKind := s.aggregation
	if aggKind == "" {
		aggKind = comet.SumAggregation
	}
	agg, err := comet.NewTextAggregation(aggKind)
	if err != nil {
		return nil, err
	}
	ix := s.index
	ix.mu.RLock()
	defer ix.mu.RUnlock()
	all := append([]string{}, s.queries...)
	for _, id := range s.nodeIDs { // lookupNodeTexts bm25_index_search.go:192-220
		if _, gone := ix.deleted[id]; gone {
			return nil, fmt.Errorf("node ID %d not found in index (deleted)", id)
		}
		toks, ok := ix.docTokens[id]
		if !ok {
			return nil, fmt.Errorf("node ID %d not found in index", id)
		}
		all = append(all, strings.Join(toks, " "))
	}
	B := len(all)
	offs := make([]int32, B+1)
	var qtok []uint32
	for b, q := range all {
		qtok = append(qtok, ix.tokenIDs(tokenize(normalize(q)), false)...)
		offs[b+1] = int32(len(qtok))
	}
	kCap := s.k
	if kCap <= 0 || kCap > 2048 {
		kCap = 2048
	}
	ids := make([]uint32, B*kCap)
	scores := make([]float32, B*kCap)
	counts := make([]int32, B)
	var qp, fp *C.uint32_t
	if len(qtok) > 0 {
		qp = (*C.uint32_t)(&qtok[0])
	}
	if len(s.documentIDs) > 0 {
		fp = (*C.uint32_t)(&s.documentIDs[0])
	}
	rc := C.comet_bm25_search(ix.h, qp, (*C.int32_t)(&offs[0]), C.int32_t(B), C.int32_t(s.k), fp, C.int32_t(len(s.documentIDs)),
		(*C.uint32_t)(&ids[0]), (*C.float)(&scores[0]), nil, (*C.int32_t)(&counts[0]), C.int32_t(kCap))
	if rc != C.COMET_OK {
		return nil, lastError(rc)
	}
	var results []comet.TextResult
	for b := 0; b < B; b++ {
		c := int(counts[b])
		if c > kCap {
			c = kCap
		}
		for i := 0; i < c; i++ {
			results = append(results, comet.TextResult{Id: ids[b*kCap+i], Score: scores[b*kCap+i]})
		}
	}
	results = agg.Aggregate(results)
	results = comet.LimitResults(results, s.k)
	results = comet.AutocutResults(results, s.cutoff)
	return results, nil
}

// ---- io.WriterTo / io.ReaderFrom: the reference's "BM25" layout (bm25_index.go:421-460, WriteTo :467, ReadFrom :640) ----
// Built from the host-side token lists (the GPU holds only ids); WriteTo flushes first like the reference.
func (ix *BM25SearchIndex) WriteTo(w io.Writer) (int64, error) {
	if err := ix.Flush(); err != nil {
		return 0, fmt.Errorf("failed to flush before serialization: %w", err)
	}
	ix.mu.RLock()
	defer ix.mu.RUnlock()
	cw := &countingWriter{w: w}
	put := func(v interface{}) error { return binary.Write(cw, binary.LittleEndian, v) }
	putStr := func(s string) error {
		if err := put(uint32(len(s))); err != nil {
			return err
		}
		_, err := cw.Write([]byte(s))
		return err
	}
	docIDs := make([]uint32, 0, len(ix.docTokens))
	total := 0
	for id, t := range ix.docTokens {
		docIDs = append(docIDs, id)
		total += len(t)
	}
	sort.Slice(docIDs, func(i, j int) bool { return docIDs[i] < docIDs[j] })
	avg := 0.0
	if len(docIDs) > 0 {
		avg = float64(total) / float64(len(docIDs))
	}
	steps := []func() error{
		func() error { _, err := cw.Write([]byte("BM25")); return err },
		func() error { return put(uint32(1)) },
		func() error { return put(uint32(len(docIDs))) },
		func() error { return put(uint32(total)) },
		func() error { return put(avg) },
		func() error { // document lengths
			if err := put(uint32(len(docIDs))); err != nil {
				return err
			}
			for _, id := range docIDs {
				if err := put(id); err != nil {
					return err
				}
				if err := put(uint32(len(ix.docTokens[id]))); err != nil {
					return err
				}
			}
			return nil
		},
		func() error { // document tokens
			if err := put(uint32(len(docIDs))); err != nil {
				return err
			}
			for _, id := range docIDs {
				if err := put(id); err != nil {
					return err
				}
				if err := put(uint32(len(ix.docTokens[id]))); err != nil {
					return err
				}
				for _, t := range ix.docTokens[id] {
					if err := putStr(t); err != nil {
						return err
					}
				}
			}
			return nil
		},
	}
	for _, f := range steps {
		if err := f(); err != nil {
			return cw.n, err
		}
	}
	// postings + term frequencies
	post := map[string]*roaring.Bitmap{}
	tf := map[string]map[uint32]int{}
	for _, id := range docIDs {
		for _, t := range ix.docTokens[id] {
			if post[t] == nil {
				post[t] = roaring.New()
				tf[t] = map[uint32]int{}
			}
			post[t].Add(id)
			tf[t][id]++
		}
	}
	terms := make([]string, 0, len(post))
	for t := range post {
		terms = append(terms, t)
	}
	sort.Strings(terms)
	if err := put(uint32(len(terms))); err != nil {
		return cw.n, err
	}
	for _, t := range terms {
		if err := putStr(t); err != nil {
			return cw.n, err
		}
		b, err := post[t].ToBytes()
		if err != nil {
			return cw.n, err
		}
		if err := put(uint32(len(b))); err != nil {
			return cw.n, err
		}
		if _, err := cw.Write(b); err != nil {
			return cw.n, err
		}
	}
	if err := put(uint32(len(terms))); err != nil {
		return cw.n, err
	}
	for _, t := range terms {
		if err := putStr(t); err != nil {
			return cw.n, err
		}
		if err := put(uint32(len(tf[t]))); err != nil {
			return cw.n, err
		}
		docs := make([]uint32, 0, len(tf[t]))
		for d := range tf[t] {
			docs = append(docs, d)
		}
		sort.Slice(docs, func(i, j int) bool { return docs[i] < docs[j] })
		for _, d := range docs {
			if err := put(d); err != nil {
				return cw.n, err
			}
			if err := put(uint32(tf[t][d])); err != nil {
				return cw.n, err
			}
		}
	}
	empty, _ := roaring.New().ToBytes()
	if err := put(uint32(len(empty))); err != nil {
		return cw.n, err
	}
	_, err := cw.Write(empty)
	return cw.n, err
}

// ReadFrom loads the reference's layout by delegating the parsing to the reference's own reader and replaying the
// documents (token lists) into the GPU index.
func (ix *BM25SearchIndex) ReadFrom(r io.Reader) (int64, error) {
	ref := comet.NewBM25SearchIndex()
	n, err := ref.ReadFrom(r)
	if err != nil {
		return n, err
	}
	// the reference keeps docTokens private; round-trip through its WithNode lookups is not possible, so parse the
	// "document tokens" section ourselves from a re-serialisation
	pr, pw := io.Pipe()
	go func() { _, e := ref.WriteTo(pw); pw.CloseWithError(e) }()
	docs, err := parseDocTokens(pr)
	if err != nil {
		return n, err
	}
	fresh, err := NewBM25SearchIndex(ix.ctx)
	if err != nil {
		return n, err
	}
	ids := make([]uint32, 0, len(docs))
	for id := range docs {
		ids = append(ids, id)
	}
	sort.Slice(ids, func(i, j int) bool { return ids[i] < ids[j] })
	for _, id := range ids {
		if err := fresh.addTokens(id, docs[id]); err != nil {
			return n, err
		}
	}
	ix.mu.Lock()
	defer ix.mu.Unlock()
	C.comet_bm25_destroy(ix.h)
	ix.h, ix.vocab, ix.docTokens, ix.deleted = fresh.h, fresh.vocab, fresh.docTokens, fresh.deleted
	fresh.h = nil
	return n, nil
}

func parseDocTokens(r io.Reader) (map[uint32][]string, error) {
	get := func(v interface{}) error { return binary.Read(r, binary.LittleEndian, v) }
	magic := make([]byte, 4)
	if _, err := io.ReadFull(r, magic); err != nil {
		return nil, err
	}
	var version, numDocs, totalTokens, n uint32
	var avg float64
	for _, p := range []interface{}{&version, &numDocs, &totalTokens, &avg, &n} {
		if err := get(p); err != nil {
			return nil, err
		}
	}
	for i := uint32(0); i < n; i++ { // document lengths
		var id, l uint32
		if err := get(&id); err != nil {
			return nil, err
		}
		if err := get(&l); err != nil {
			return nil, err
		}
	}
	if err := get(&n); err != nil {
		return nil, err
	}
	docs := make(map[uint32][]string, n)
	for i := uint32(0); i < n; i++ {
		var id, cnt uint32
		if err := get(&id); err != nil {
			return nil, err
		}
		if err := get(&cnt); err != nil {
			return nil, err
		}
		toks := make([]string, cnt)
		for j := range toks {
			var l uint32
			if err := get(&l); err != nil {
				return nil, err
			}
			b := make([]byte, l)
			if _, err := io.ReadFull(r, b); err != nil {
				return nil, err
			}
			toks[j] = string(b)
		}
		docs[id] = toks
	}
	_, _ = io.Copy(io.Discard, r)
	return docs, nil
}

type countingWriter struct {
	w io.Writer
	n int64
}

func (c *countingWriter) Write(p []byte) (int, error) {
	n, err := c.w.Write(p)
	c.n += int64(n)
	return n, err
}
