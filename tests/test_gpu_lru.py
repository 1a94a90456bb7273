"""GPU parity for stage 1: the HIP path (through the C ABI) must equal the CPU oracle BIT FOR BIT
(retrieved top-K indices and scores, last-position hidden state, full score rows), and match the
reference's goldens at fp tolerance."""
import json
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def env(golden_dir):
    from llamarec_amd.lru import LRURec
    from oracle import lru_oracle as O

    z = np.load(os.path.join(golden_dir, "lru_v300.npz"))
    sd = {k[3:]: z[k] for k in z.files if k.startswith("sd/")}
    return z, sd, LRURec.from_state_dict(sd), O.LruOracle(sd), O


def _bits(a):
    return np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)


@pytest.mark.parametrize("case", ["L50", "L7", "L64", "L200"])
def test_encode_and_scores_bit_exact(env, case):
    z, _, model, orc, _ = env
    ids = z[f"ids/{case}"]
    q = model.encode_last(ids).cpu().numpy()
    assert np.array_equal(_bits(q), _bits(orc.encode_last(ids)))
    for excl in (False, True):
        s = model.scores_last(ids, excl).cpu().numpy()
        assert np.array_equal(_bits(s), _bits(orc.scores_last(ids, excl)))
    # reference idiom + golden tolerance
    ref = z[f"scores_last/{case}"]
    got = model(torch.from_numpy(ids))[:, -1, :].cpu().numpy()
    assert np.abs(got - ref).max() < 2e-5


@pytest.mark.parametrize("case,K,excl", [("L50", 20, True), ("L50", 50, True), ("L200", 50, True),
                                         ("L7", 20, False), ("L64", 64, True), ("L50", 1, True)])
def test_topk_bit_exact(env, case, K, excl):
    z, _, model, orc, _ = env
    ids = z[f"ids/{case}"]
    idx, sc = model.retrieve_topk(ids, K, excl)
    oi, os_ = orc.retrieve_topk(ids, K, excl)
    assert np.array_equal(idx.cpu().numpy(), oi)
    assert np.array_equal(_bits(sc.cpu().numpy()), _bits(os_))


def test_generate_candidates_golden(env, golden_dir):
    """Ordered top-50 lists == the reference's retrieved.pkl content (trainer/lru.py:113-115)."""
    _, _, model, _, _ = env
    g = json.load(open(os.path.join(golden_dir, "candidates.json")))
    zin = np.load(os.path.join(golden_dir, "candidates_inputs.npz"))
    idx, _ = model.retrieve_topk(zin["test_ids"], 50, True)
    assert idx.cpu().numpy().tolist() == g["retrieved"]["test_probs"]


def _big_case(V, B, L, seed, nb=2):
    from llamarec_amd.lru import init_lru_state_dict

    rng = np.random.default_rng(seed)
    sd = init_lru_state_dict(V, seed, nb)
    ids = np.zeros((B, L), np.int64)
    for b in range(B):
        n = int(rng.integers(1, L + 1))
        ids[b, L - n:] = rng.integers(1, V + 1, size=n)
    return sd, ids


@pytest.mark.parametrize("V,B,L,K", [(12086, 300, 50, 50), (3650, 37, 200, 20), (40000, 5, 50, 50),
                                     (33, 9, 50, 50), (31, 130, 8, 20)])
def test_topk_bit_exact_larger_shapes(V, B, L, K):
    """Beauty / ML-100k shapes, a small batch over many item chunks, and catalogs smaller than
    K + history (masked -1e9 entries then surface in the list, ties by id)."""
    from llamarec_amd.lru import LRURec
    from oracle import lru_oracle as O

    sd, ids = _big_case(V, B, L, seed=V + B)
    model, orc = LRURec.from_state_dict(sd), O.LruOracle(sd)
    idx, sc = model.retrieve_topk(ids, K, True)
    oi, os_ = orc.retrieve_topk(ids, K, True)
    assert np.array_equal(idx.cpu().numpy(), oi)
    assert np.array_equal(_bits(sc.cpu().numpy()), _bits(os_))
    idx2, _ = model.retrieve_topk(ids, K, True)  # determinism
    assert torch.equal(idx, idx2)


def test_single_block_and_three_blocks():
    from llamarec_amd.lru import LRURec
    from oracle import lru_oracle as O

    for nb in (1, 3):
        sd, ids = _big_case(500, 20, 30, seed=nb, nb=nb)
        model, orc = LRURec.from_state_dict(sd), O.LruOracle(sd)
        assert np.array_equal(_bits(model.encode_last(ids).cpu().numpy()), _bits(orc.encode_last(ids)))


def test_metrics_histogram(env, golden_dir):
    from llamarec_amd import metrics as M

    _, _, _, _, O = env
    z = np.load(os.path.join(golden_dir, "metrics.npz"))
    g = json.load(open(os.path.join(golden_dir, "metrics.json")))
    ranked = torch.from_numpy(z["ranked"].astype(np.int32)).cuda()
    labels = torch.from_numpy(z["labels"]).cuda()
    got = M.absolute_recall_mrr_ndcg_for_ks(ranked, labels, g["ks"], preprocessed=True)
    assert list(got.keys()) == list(g["full"].keys())
    for k, v in g["full"].items():
        assert abs(got[k] - v) < 1e-6
    sums = M.metric_sums_from_histogram(M.rank_histogram(ranked, labels), g["ks"])
    assert np.allclose(sums, O.rank_metric_sums(z["ranked"], z["labels"], g["ks"]), rtol=0, atol=1e-12)
    # reranker: raw [N,20] scores
    got = M.absolute_recall_mrr_ndcg_for_ks(torch.from_numpy(z["s20"]).cuda(), torch.from_numpy(z["l20"]).cuda(),
                                            g["rerank_ks"])
    for k, v in g["rerank"].items():
        assert abs(got[k] - v) < 1e-6
    oi, _ = O.topk(z["s20"], 20)
    assert np.array_equal(M.rank_classes(torch.from_numpy(z["s20"]).cuda()).cpu().numpy(), oi)


@pytest.mark.parametrize("name,U", [("ml-100k", 610), ("beauty", 22332), ("games", 15264)])
def test_baseline_shapes_all_users_vs_oracle(name, U):
    """All users of a BASELINE config in ONE call (many user tiles / chunks, batched encoder over ~10^5 rows): EVERY
    user's ordered top-50 equals the oracle's bit for bit (ids and score bits; the C oracle ranks 22 k users in seconds)."""
    from llamarec_amd.lru import LRURec, init_lru_state_dict
    from llamarec_amd.synth import WORKLOADS, synth_users
    from oracle import lru_oracle as O

    w = WORKLOADS[name]
    hist, labels, n, T = synth_users(name, U)
    sd = init_lru_state_dict(w["V"], seed=42)
    idx, sc = LRURec.from_state_dict(sd).retrieve_topk(torch.from_numpy(hist).cuda(), 50, True)
    oi, osc = O.LruOracle(sd).retrieve_topk(hist, 50, True)
    assert np.array_equal(idx.cpu().numpy(), oi)
    assert np.array_equal(sc.cpu().numpy().view(np.uint32), osc.view(np.uint32))


def test_full_size_catalog_properties():
    """BASELINE configs[4] size (1M-item table, L = 200), where the CPU oracle is too slow to rank every user:
    size-independent properties of the fused retrieve, plus the oracle on a sample of users.
      * scores come back sorted (desc; ties -> lower id) and equal the materialised scores at the returned ids
      * nothing from the user's history and no pad id is returned; ids are distinct and in range
      * top-20 is the prefix of top-50 (what generate_candidates relies on, trainer/lru.py:82-84,113-115)
      * a user's list does not depend on which other users share the call (chunk geometry changes with B)."""
    from llamarec_amd.lru import LRURec, init_lru_state_dict
    from llamarec_amd.synth import synth_users
    from oracle import lru_oracle as O

    V, U = 1_000_000, 300
    hist, labels, n, T = synth_users("synth-1m", U)
    sd = init_lru_state_dict(V, seed=42)
    model = LRURec.from_state_dict(sd)
    ids = torch.from_numpy(hist).cuda()
    i50, s50 = model.retrieve_topk(ids, 50, True)
    i20, s20 = model.retrieve_topk(ids, 20, True)
    i50n, s50n = i50.cpu().numpy(), s50.cpu().numpy()
    assert np.array_equal(i20.cpu().numpy(), i50n[:, :20]) and np.array_equal(s20.cpu().numpy(), s50n[:, :20])
    assert (np.diff(s50n, axis=1) <= 0).all()
    ties = np.diff(s50n, axis=1) == 0
    assert (np.diff(i50n, axis=1)[ties] > 0).all()
    assert i50n.min() >= 1 and i50n.max() <= V
    for u in range(U):
        assert len(set(i50n[u])) == 50 and not (set(i50n[u]) & set(hist[u]))
    # fused scores == materialised scores at the same ids (different kernels, same arithmetic)
    sub = slice(0, 6)
    full = model.scores_last(ids[sub], exclude_history=True)
    assert torch.equal(torch.gather(full, 1, i50[sub].long()), s50[sub])
    assert torch.equal(full.topk(50, dim=1).values, s50[sub])
    # batch composition must not matter
    alone_i, alone_s = model.retrieve_topk(ids[7:8], 50, True)
    assert torch.equal(alone_i[0], i50[7]) and torch.equal(alone_s[0], s50[7])
    half_i, _ = model.retrieve_topk(ids[100:229], 50, True)
    assert torch.equal(half_i, i50[100:229])
    # and the oracle on four users (about a second each on the CPU): since round 3 the 1 M-item catalog takes the bf16
    # bound -> candidates -> exact rescoring path with one maximum per GROUP of 16 tiles (lru_topk.hip, bound_group_shift)
    pick = [3, 211, 77, 298]
    oi, osc = O.LruOracle(sd).retrieve_topk(hist[pick], 50, True)
    assert np.array_equal(oi, i50n[pick]) and np.array_equal(osc, s50n[pick])


def test_many_users_on_the_full_size_catalog_keep_the_candidate_path():
    """at 1 M items the candidate pass's chunk count used to shrink with the number of user tiles
    (chunks = 1024 / n_ut): 16 384 users got 32 chunks of ~1 000 tiles, ~12 expected candidates per user and chunk against
    24 LDS slots, so some list overflowed on essentially every call and the exact full pass redid it -- correct results, a
    performance cliff above the measured 4 096-user point. Chunks are now at most 512 tiles whatever the batch
    (lr_bf16_max_chunk_tiles); lr_lru_topk_path reports which path a call took. 16 384 users (L = 50 keeps the encoder
    short): path 1, and a user's list equals the one it gets in a 300-user call (and, for four users, the oracle's)."""
    from llamarec_amd.lru import LRURec, init_lru_state_dict
    from llamarec_amd.synth import synth_users
    from oracle import lru_oracle as O

    V, U, L = 1_000_000, 16384, 50
    hist = synth_users("synth-1m", U)[0][:, -L:]
    sd = init_lru_state_dict(V, seed=42)
    model = LRURec.from_state_dict(sd)
    ids = torch.from_numpy(np.ascontiguousarray(hist)).cuda()
    i50, s50 = model.retrieve_topk(ids, 50, True)
    assert model.last_topk_path(U, L, 50, True) == 1          # no candidate list overflowed
    small_i, small_s = model.retrieve_topk(ids[5000:5300], 50, True)
    assert model.last_topk_path(300, L, 50, True) == 1
    assert torch.equal(small_i, i50[5000:5300]) and torch.equal(small_s, s50[5000:5300])
    pick = [0, 4097, 9999, 16383]
    oi, osc = O.LruOracle(sd).retrieve_topk(np.ascontiguousarray(hist[pick]), 50, True)
    assert np.array_equal(oi, i50.cpu().numpy()[pick]) and np.array_equal(osc, s50.cpu().numpy()[pick])
    # shapes the bound does not serve report path 0 (ML-100k: 200-id histories against 115 tiles)
    small = LRURec.from_state_dict(init_lru_state_dict(3650, seed=1))
    h2 = torch.from_numpy(synth_users("ml-100k", 32)[0]).cuda()
    small.retrieve_topk(h2, 20, True)
    assert small.last_topk_path(32, h2.shape[1], 20, True) == 0


@pytest.mark.parametrize("V,L", [(100_000, 50), (70_003, 200), (262_144, 20)])
def test_grouped_bound_catalogs_vs_oracle(V, L):
    """Catalogs beyond 65 536 items (2 048 tiles) keep one approximate maximum per group of 4 / 8 / 16 tiles; the proof of
    the bound is unchanged (the R-th largest GROUP maximum certifies R different items at or above it), so the ordered
    top-50 and its score bits must equal the C oracle's for every user: 3 125 tiles -> groups of 4 (the last group
    ragged), 2 188 -> groups of 4 with L = 200 masked ids, 8 193 tiles -> groups of 8 with one tile in the last group.
    Includes users whose whole history sits in their own top-50 and a block of exactly tied items."""
    from llamarec_amd.lru import LRURec, init_lru_state_dict
    from oracle import lru_oracle as O

    B = 96
    rng = np.random.default_rng(V)
    sd = init_lru_state_dict(V, seed=V % 1000)
    ids = np.zeros((B, L), np.int64)
    for b in range(B):
        n = int(rng.integers(1, L + 1))
        ids[b, L - n:] = rng.choice(V, size=n, replace=False) + 1
    model = LRURec.from_state_dict(sd)
    orc = O.LruOracle(sd)
    first, _ = orc.retrieve_topk(ids, 50, False)
    for b in range(0, B, 7):                         # history = the user's own unmasked top-(up to L) items
        m = min(L, 50)
        ids[b] = 0
        ids[b, L - m:] = first[b, :m]
    got_i, got_s = model.retrieve_topk(torch.from_numpy(ids).cuda(), 50, True)
    oi, osc = orc.retrieve_topk(ids, 50, True)
    assert np.array_equal(got_i.cpu().numpy(), oi)
    assert np.array_equal(got_s.cpu().numpy(), osc)
    # validation mode (no masking) and a small K through the same path
    got_i, got_s = model.retrieve_topk(torch.from_numpy(ids).cuda(), 7, False)
    oi, osc = orc.retrieve_topk(ids, 7, False)
    assert np.array_equal(got_i.cpu().numpy(), oi) and np.array_equal(got_s.cpu().numpy(), osc)


@pytest.mark.parametrize("V,B", [(2047, 1), (2047, 513), (2079, 3), (2271, 513), (4097, 64), (65535, 2), (65537, 2)])
def test_bf16_stage_ring_boundaries(V, B):
    """The two bf16 passes stream the packed table through an LDS ring, 8 (bound) and 4 (candidates) tiles per stage, in
    chunks of whole 4-tile groups; a workgroup is 512 users. Catalog sizes on every edge of that geometry -- 64 tiles (the
    smallest catalog the bound serves), 65 and 71 tiles (ragged last stage, ragged last group of 4, a last tile with one /
    thirty-one padding rows scoring -inf), 129 tiles, and 2 048 / 2 049 tiles (the switch from one maximum per tile to one
    per group of 4) -- with one user, a few users, and one user more than a workgroup: ordered top-50 and score bits vs the
    C oracle for every user, on the candidate path (lr_lru_topk_path = 1)."""
    from llamarec_amd.lru import LRURec, init_lru_state_dict
    from oracle import lru_oracle as O

    L, K = 10, 50
    rng = np.random.default_rng(V * 7 + B)
    sd = init_lru_state_dict(V, seed=V % 97)
    sd["model.bias"] = (rng.standard_normal(V + 1) * 0.05).astype(np.float32)   # non-trivial biases (the MFMA's C operand)
    ids = np.zeros((B, L), np.int64)
    for b in range(B):
        n = int(rng.integers(1, L + 1))
        ids[b, L - n:] = rng.integers(1, V + 1, size=n)
    ids[0, -3:] = (V, V - 1, 1)                                              # the catalog's last items are history for user 0
    model, orc = LRURec.from_state_dict(sd), O.LruOracle(sd)
    idx, sc = model.retrieve_topk(ids, K, True)
    assert model.last_topk_path(B, L, K, True) == 1
    pick = np.unique(np.concatenate([np.arange(min(B, 6)), np.arange(max(0, B - 3), B), np.arange(0, B, 97)]))
    oi, os_ = orc.retrieve_topk(ids[pick], K, True)
    assert np.array_equal(idx.cpu().numpy()[pick], oi)
    assert np.array_equal(_bits(sc.cpu().numpy()[pick]), _bits(os_))


@pytest.mark.parametrize("variant", ["norms_and_bias", "history_is_the_top", "tie_blocks", "all_equal", "no_exclude_k7"])
def test_candidate_path_adversarial(variant):
    """Catalogs of 64..2048 tiles take the bf16 bound + candidate + exact-rescore path (lru_topk.hip): its output must
    still be the oracle's ordered top-K BIT FOR BIT when the proven error bound is stressed --
      norms_and_bias      item rows scaled by 0.05..20 and biases ~ N(0, 1): a loose delta, many candidates
      history_is_the_top  every user's history = its own unmasked top-50 (full L): the masked-id count R = K + 51 is needed
      tie_blocks          100 items with identical rows and biases (exact score ties, broken by id) around the top
      all_equal           every item identical: every approximate score passes, the candidate lists overflow and the
                          device-side flag must hand the call to the exact full pass
      no_exclude_k7       validation mode (no masking, the pad id 0 stays eligible) with a small K."""
    from llamarec_amd.lru import LRURec, init_lru_state_dict
    from oracle import lru_oracle as O

    V, B, L = 4000, 70, 50          # 126 tiles: seeded (K + L + 1 <= tiles)
    rng = np.random.default_rng(hash(variant) % 2 ** 31)
    sd = init_lru_state_dict(V, seed=5)
    E = sd["embedding.token.weight"].copy()
    bias = sd["model.bias"].copy()
    K, excl = 50, True
    if variant == "norms_and_bias":
        E *= rng.uniform(0.05, 20.0, size=(V + 1, 1)).astype(np.float32)
        bias = rng.standard_normal(V + 1).astype(np.float32)
    elif variant == "tie_blocks":
        E[100:200] = E[100] * np.float32(6.0)        # large norm: these rows are near the top for many users
        bias[100:200] = np.float32(0.25)
    elif variant == "all_equal":
        E[:] = E[7]
        bias[:] = np.float32(0.5)
    elif variant == "no_exclude_k7":
        K, excl = 7, False
        E[0] *= np.float32(30.0)                      # the pad row is a strong candidate when it is not masked
    sd["embedding.token.weight"], sd["model.bias"] = E, bias
    ids = np.zeros((B, L), np.int64)
    for b in range(B):
        n = int(rng.integers(1, L + 1))
        ids[b, L - n:] = rng.integers(1, V + 1, size=n)
    model, orc = LRURec.from_state_dict(sd), O.LruOracle(sd)
    if variant == "history_is_the_top":
        top_unmasked, _ = orc.retrieve_topk(ids, 50, False)
        ids = np.where(top_unmasked == 0, 1, top_unmasked).astype(np.int64)   # (history ids: 0 is the pad)
    idx, sc = model.retrieve_topk(ids, K, excl)
    # the overflow hand-over really happened where it was provoked, and only there (lr_lru_topk_path)
    # (no_exclude_k7 scales ONE row by 30: the bound's delta is proportional to the table's largest row norm, so its
    # candidate threshold sinks below most of the catalog and the lists may overflow too -- either path is legitimate)
    path = model.last_topk_path(B, L, K, excl)
    assert path == 2 if variant == "all_equal" else path in ((1, 2) if variant == "no_exclude_k7" else (1,)), path
    oi, os_ = orc.retrieve_topk(ids, K, excl)
    assert np.array_equal(idx.cpu().numpy(), oi)
    assert np.array_equal(_bits(sc.cpu().numpy()), _bits(os_))
    idx2, sc2 = model.retrieve_topk(ids[5:6], K, excl)      # alone in the call: same list
    assert np.array_equal(idx2.cpu().numpy(), oi[5:6]) and np.array_equal(_bits(sc2.cpu().numpy()), _bits(os_[5:6]))


def test_pipelined_and_plain_encoder_kernels_are_bit_identical():
    """em_pipe_kernel (two tiles in flight, phase B weights in registers, four-segment recurrence on the A waves) against
    em_layer_kernel (one tile at a time, eight segments) for the LRU layer: the same chains in the same order, so the last
    hidden state must be BIT-IDENTICAL -- on short users (many segments per tile), long users (one user per tile: the
    recurrence's fast path and the carry across tiles), a mix, three blocks (two pipelined layers), and a batch small
    enough that most workgroups own a single tile."""
    from llamarec_amd.lru import LRURec, init_lru_state_dict

    rng = np.random.default_rng(21)
    for V, L, B, nb in ((900, 257, 3000, 2), (900, 50, 7000, 3), (500, 200, 40, 2)):
        sd = init_lru_state_dict(V, seed=V + nb, num_blocks=nb)
        ids = np.zeros((B, L), np.int64)
        n = rng.integers(1, L + 1, size=B)
        short = rng.random(B) < 0.5
        n[short] = rng.integers(1, 10, size=int(short.sum()))
        n[:3] = (L, 1, min(L, 65))
        for b in range(B):
            ids[b, L - n[b]:] = rng.integers(1, V + 1, size=n[b])
        model = LRURec.from_state_dict(sd)
        q_pipe = model.encode_last(ids).cpu().numpy()
        model.set_encoder_pipeline(False)
        q_plain = model.encode_last(ids).cpu().numpy()
        model.set_encoder_pipeline(True)
        assert np.isfinite(q_pipe).all()
        assert np.array_equal(_bits(q_pipe), _bits(q_plain)), (V, L, B, nb)


def test_encoder_chunks_segments_and_batch_independence():
    """The batched encoder beyond one chunk of users (2^21 row slots: 8 160 users at L = 257), with users of 1 .. 257 live
    rows so that super tiles hold anything from one slice of a long user (carry across tiles, no segment cut) to a
    dozen short users (four-way segment cut): a spread sample equals the oracle bit for bit, and a user's state does not
    depend on which other users share the call (row ranges, tiles and segments all move with the batch)."""
    from llamarec_amd.lru import LRURec, init_lru_state_dict
    from oracle import lru_oracle as O

    V, L, B = 1500, 257, 9000
    rng = np.random.default_rng(11)
    sd = init_lru_state_dict(V, seed=9)
    ids = np.zeros((B, L), np.int64)
    n = rng.integers(1, L + 1, size=B)
    short = rng.random(B) < 0.6
    n[short] = rng.integers(1, 12, size=int(short.sum()))
    n[:4] = (L, 1, 64, 65)
    for b in range(B):
        ids[b, L - n[b]:] = rng.integers(1, V + 1, size=n[b])
    model = LRURec.from_state_dict(sd)
    q = model.encode_last(ids).cpu().numpy()
    sample = np.concatenate([np.arange(0, 6), np.arange(8150, 8170, 4), np.arange(B - 3, B)])
    want = O.LruOracle(sd).encode_last(ids[sample])
    assert np.array_equal(_bits(q[sample]), _bits(want))
    sub = np.concatenate([np.arange(100, 400), np.arange(8000, 8300)])
    q_sub = model.encode_last(ids[sub]).cpu().numpy()
    assert np.array_equal(_bits(q_sub), _bits(q[sub]))
