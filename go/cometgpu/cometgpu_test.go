package cometgpu

import (
	"bytes"
	"testing"

	comet "github.com/wizenheimer/comet"
)

// The GPU index and the pure-Go index must answer identically on the same inputs (Flat L2Squared: ids / ranks bit-exact;
// every other path: see the backend's Python / C test-suite, which compares against a CPU restatement of these loops).
func TestFlatMatchesReference(t *testing.T) {
	ctx, err := NewContext(0)
	if err != nil {
		t.Skipf("no MI355X: %v", err)
	}
	defer ctx.Close()
	gpu, err := NewFlatIndex(ctx, 3, comet.L2Squared)
	if err != nil {
		t.Fatal(err)
	}
	ref, _ := comet.NewFlatIndex(3, comet.L2Squared)
	for i := 0; i < 100; i++ { // flat_index_search_test.go:392-420 shape
		v := []float32{float32(i), 0, 0}
		if err := gpu.Add(*comet.NewVectorNodeWithID(uint32(i+1), append([]float32{}, v...))); err != nil {
			t.Fatal(err)
		}
		ref.Add(*comet.NewVectorNodeWithID(uint32(i+1), v))
	}
	q := []float32{7.25, 0, 0}
	got, err := gpu.NewSearch().WithQuery(q).WithK(10).Execute()
	if err != nil {
		t.Fatal(err)
	}
	want, _ := ref.NewSearch().WithQuery(q).WithK(10).Execute()
	if len(got) != len(want) {
		t.Fatalf("len %d vs %d", len(got), len(want))
	}
	for i := range got {
		if got[i].GetId() != want[i].GetId() || got[i].Score != want[i].Score {
			t.Errorf("rank %d: gpu (%d, %v) vs reference (%d, %v)", i, got[i].GetId(), got[i].Score, want[i].GetId(), want[i].Score)
		}
	}
}

// A GPU index is a comet.VectorIndex: it serialises to the reference's own format and drops into the hybrid index.
func TestDropInAndOnDiskFormat(t *testing.T) {
	ctx, err := NewContext(0)
	if err != nil {
		t.Skipf("no MI355X: %v", err)
	}
	defer ctx.Close()
	gpu, _ := NewFlatIndex(ctx, 4, comet.Cosine)
	gpu.Add(*comet.NewVectorNodeWithID(1, []float32{1, 2, 3, 4}))
	gpu.Add(*comet.NewVectorNodeWithID(2, []float32{4, 3, 2, 1}))
	var buf bytes.Buffer
	if _, err := gpu.WriteTo(&buf); err != nil {
		t.Fatal(err)
	}
	ref, _ := comet.NewFlatIndex(4, comet.Cosine)
	if _, err := ref.ReadFrom(bytes.NewReader(buf.Bytes())); err != nil { // the pure-Go index reads what the GPU wrote
		t.Fatal(err)
	}
	var again bytes.Buffer
	ref.WriteTo(&again)
	if !bytes.Equal(buf.Bytes(), again.Bytes()) {
		t.Error("GPU and reference serialisations differ")
	}
	txt, _ := NewBM25SearchIndex(ctx)
	hybrid := comet.NewHybridSearchIndex(gpu, txt, comet.NewRoaringMetadataIndex())
	if hybrid == nil {
		t.Fatal("hybrid index rejected the GPU indexes")
	}
}
