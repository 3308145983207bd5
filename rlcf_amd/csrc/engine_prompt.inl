// Part of engine.hip (one translation unit: #include'd there): the prompt-tuning step — sparse class passes, one test sample
// (engine_tta_sample: TPT/tpt_cls_rl.py:47-79 + the harness body :251-268), B test samples per tower pass (engine_tta_batch).
// Sparse backward layout for n_e = n_sel*K sampled (view, class) pairs.
static int sparse_ensure(rlcf_engine* e, int n_e_per_group, hipStream_t st, int groups = 1) {
    if (n_e_per_group <= e->sp_max_e && groups <= e->sp_groups) return RLCF_OK;
    ClipModel& m = e->model[RLCF_STUDENT];
    const TextLayout& L = e->lay[0];
    const int Wt = m.cfg.text_width, D = m.cfg.embed_dim;
    groups = std::max(groups, e->sp_groups);
    n_e_per_group = std::max(n_e_per_group, e->sp_max_e);
    const int T = groups * (L.pre_rows + n_e_per_group * L.lmax);
    const int n_e = groups * n_e_per_group;
    TRY(e->sp_seqs.ensure((size_t)(n_e + groups) * sizeof(rlcf_seq))); TRY(e->sp_eot_rows.ensure(n_e * sizeof(int32_t)));
    TRY(e->sp_row_src.ensure((size_t)T * sizeof(int32_t)));
    std::vector<int32_t> list;
    if (L.pre_rows > 0) for (int j = 0; j < L.n_ctx; ++j) list.push_back(1 + j);
    else for (int k = 0; k < n_e_per_group; ++k) for (int j = 0; j < L.n_ctx; ++j) list.push_back(k * L.lmax + 1 + j);
    TRY(upload(e->sp_ctx_rows_list, list, st));
    TRY(e->sp_dtxt.ensure((size_t)n_e * D * sizeof(float))); TRY(e->sp_txt.ensure((size_t)n_e * D * sizeof(float)));
    TRY(e->sp_inv_norm.ensure(n_e * sizeof(float))); TRY(e->sp_eot_x.ensure((size_t)n_e * Wt * sizeof(float)));
    TRY(e->sp_eot_ln.ensure((size_t)n_e * Wt * sizeof(float))); TRY(e->sp_u.ensure((size_t)n_e * D * sizeof(float)));
    TRY(e->sp_du.ensure((size_t)std::max(n_e, L.C) * D * sizeof(float))); TRY(e->sp_dxe.ensure((size_t)std::max(n_e, L.C) * Wt * sizeof(float)));
    TRY(tower_ensure(e->st, T, Wt, st));
    TRY(tower_ensure_saved(e->st, T, Wt, m.cfg.text_layers, st));
    TRY(bwd_ensure(e, T, Wt));
    e->sp_max_e = n_e_per_group; e->sp_T = T; e->sp_groups = groups;
    return RLCF_OK;
}

// the two halves of the sparse pass: the forward over the sampled (view, class) pairs needs only their class indices (top-K of the
// student's own logits) — the one-image call runs it next to the reward models' tower pass —, the backward needs the rewards
static TextPassIO sparse_io(rlcf_engine* e, int n_e) {
    const TextLayout& L = e->lay[0];
    TextPassIO io{};
    io.seqs = e->sp_seqs.as<rlcf_seq>(); io.n_seq = n_e + (L.pre_rows > 0 ? 1 : 0); io.max_q_len = L.max_q_len; io.T = L.pre_rows + n_e * L.lmax;
    io.n_cls = n_e;
    io.attn_pairs = (long)(n_e * (L.mean_len * (L.pre_rows + (L.mean_len + 1) * 0.5)));
    io.eot_rows = e->sp_eot_rows.as<int32_t>(); io.row_src = e->sp_row_src.as<int32_t>();
    io.eot_x = e->sp_eot_x.as<float>(); io.eot_ln = e->sp_eot_ln.as<float>(); io.u = e->sp_u.as<float>();
    io.inv_norm = e->sp_inv_norm.as<float>(); io.txt = e->sp_txt.as<float>();
    io.ctx_row_tab = L.ctx_general ? L.ctx_row.as<int32_t>() : nullptr;
    return io;
}
static int sparse_forward(rlcf_engine* e, const float* ctx, const int32_t* cls, int n_e, hipStream_t st) {
    ClipModel& m = e->model[RLCF_STUDENT];
    const TextLayout& L = e->lay[0];
    TRY(launch_build_sparse_layout(cls, 1, n_e, L.class_start.as<int32_t>(), L.class_len.as<int32_t>(), L.class_eot_off.as<int32_t>(),
                                   L.lmax, L.pre_rows, e->sp_seqs.as<rlcf_seq>(), e->sp_eot_rows.as<int32_t>(),
                                   e->sp_row_src.as<int32_t>(), st));
    return text_forward(e, m, L, e->st, ctx, sparse_io(e, n_e), true, st);
}
static int sparse_backward_only(rlcf_engine* e, const float* sel_feat, const int32_t* cls, int n_e, int K, const float* dlogits, float* dctx,
                                hipStream_t st) {
    ClipModel& m = e->model[RLCF_STUDENT];
    const TextLayout& L = e->lay[0];
    const int D = m.cfg.embed_dim;
    TRY(launch_dtxt_sparse(dlogits, cls, sel_feat, n_e, K, L.C, D, m.logit_scale_exp, e->sp_dtxt.as<float>(), st));
    return text_backward(e, m, e->st, sparse_io(e, n_e), L.max_keys, e->sp_dtxt.as<float>(), e->sp_du.as<float>(), e->sp_dxe.as<float>(),
                         e->sp_ctx_rows_list.as<int32_t>(), L.pre_rows > 0 ? 1 : n_e, L.n_ctx, dctx, st);
}
static int sparse_backward(rlcf_engine* e, const float* ctx, const float* sel_feat, const int32_t* cls, int n_e, int K,
                           const float* dlogits, float* dctx, hipStream_t st) {
    TRY(sparse_forward(e, ctx, cls, n_e, st));
    return sparse_backward_only(e, sel_feat, cls, n_e, K, dlogits, dctx, st);
}

// ------------------------------------------------------------------ one test sample
// set_image_features of every reward model on the selected views (clip_reward.py:59-61,130-137,259-270); the optional
// output is the per-model blocks [rows, Dr_m] one after another.
static int reward_encode(rlcf_engine* e, int rows, int in_res, float* out_concat, hipStream_t st) {
    for (int m = 0; m < e->n_rewards; ++m) {
        const int Dr = e->model[RLCF_REWARD + m].cfg.embed_dim;
        TRY(engine_encode_image(e, RLCF_REWARD + m, e->views_sel.as<float>(), rows, e->rimg[m].as<float>(), st, in_res));
        if (out_concat) {
            RLCF_HIP_CHECK(hipMemcpyAsync(out_concat, e->rimg[m].p, (size_t)rows * Dr * sizeof(float), hipMemcpyDeviceToDevice, st));
            out_concat += (size_t)rows * Dr;
        }
    }
    return RLCF_OK;
}
static RewardBank reward_bank(const rlcf_engine* e) {
    RewardBank b{};
    b.n = e->n_rewards;
    for (int m = 0; m < e->n_rewards; ++m) {
        b.class_feat[m] = e->reward_cls[m].as<float>(); b.reward_img[m] = e->rimg[m].as<float>();
        b.Dr[m] = e->model[RLCF_REWARD + m].cfg.embed_dim;
        b.mix[m] = e->reward_mean ? 1.f : e->reward_mix[m];
    }
    b.post_div = e->reward_mean ? (float)e->n_rewards : 1.f;
    return b;
}

// Harness body TPT/tpt_cls_rl.py:251-262 around test_time_tuning (:47-79).
#define COPY_OUT(dst, src, bytes) do { if (dst) RLCF_HIP_CHECK(hipMemcpyAsync((dst), (src), (bytes), hipMemcpyDeviceToDevice, st)); } while (0)
int engine_tta_sample(rlcf_engine* e, const float* views, int N, const rlcf_tta_args* a, const rlcf_tta_out* out, hipStream_t st) {
    ClipModel& s = e->model[RLCF_STUDENT];
    if (e->C <= 0 || e->image_bank || e->n_rewards <= 0) { rlcf_set_error("class bank / reward model not set"); return RLCF_ERR_STATE; }
    if (e->n_ctx <= 0) { rlcf_set_error("prompt tuning needs a class bank with learnable context rows (n_ctx > 0)"); return RLCF_ERR_STATE; }
    RLCF_ARG_CHECK(N > 0 && N <= e->max_views && a && a->tta_steps >= 0 && a->sample_k > 0 && a->sample_k <= 32);
    const int C = e->C, K = a->sample_k, D = s.cfg.embed_dim, Wt = s.cfg.text_width, n_ctx = e->n_ctx;
    const int n_sel = n_selected(a, N);                       // int() truncation, tpt_cls_rl.py:34
    RLCF_ARG_CHECK(K <= C);
    if (a->tta_steps > 0 && n_sel <= 0) { rlcf_set_error("int(N*selection_p) == 0 views selected (N=%d, p=%g)", N, a->selection_p); return RLCF_ERR_ARG; }
    const size_t cb = (size_t)n_ctx * Wt * sizeof(float);
    const size_t img_elems = (size_t)3 * s.cfg.image_resolution * s.cfg.image_resolution;
    const rlcf_tta_out none{};
    if (!out) out = &none;
    const bool sparse_ok = a->sparse_backward && (a->flags & RLCF_F_REWARD_PROCESS) && !(a->flags & RLCF_F_PROCESS_BATCH) &&
                           !(a->flags & RLCF_F_MIN_ENTROPY) && K > 1;
    const int n_e = n_sel * K;
    if (a->tta_steps > 0) {
        if (sparse_ok) TRY(sparse_ensure(e, n_e, st));
        else {
            TRY(tower_ensure_saved(e->tt, e->lay[0].T, Wt, s.cfg.text_layers, st));
            TRY(bwd_ensure(e, e->lay[0].T, Wt));
            TRY(e->sp_du.ensure((size_t)C * D * sizeof(float))); TRY(e->sp_dxe.ensure((size_t)C * Wt * sizeof(float)));
        }
    }
    e->last_flops = 0.0;
    float* ctx = e->ctx.as<float>();
    // model.reset() + optimizer.load_state_dict(optim_state): custom_clip.py:161-164, tpt_cls_rl.py:251-255
    RLCF_HIP_CHECK(hipMemcpyAsync(ctx, a->ctx_in ? (const void*)a->ctx_in : e->ctx_init.p, cb, hipMemcpyDeviceToDevice, st));
    RLCF_HIP_CHECK(hipMemsetAsync(e->adam_m.p, 0, cb, st));
    RLCF_HIP_CHECK(hipMemsetAsync(e->adam_v.p, 0, cb, st));
    // student image features of all N views: computed once (the image tower is frozen, custom_clip.py:325-327)
    TRY(engine_encode_image(e, RLCF_STUDENT, views, N, e->img_feat.as<float>(), st));
    TextPassIO io = full_io(e, e->lay[0]);
    static int no_overlap = -1;                              // RLCF_NO_OVERLAP=1: everything on the caller's stream (benchmarks)
    if (no_overlap < 0) { const char* ev = getenv("RLCF_NO_OVERLAP"); no_overlap = ev ? atoi(ev) : 0; }
    bool vit_rewards = true;
    for (int m = 0; m < e->n_rewards; ++m) vit_rewards = vit_rewards && !is_resnet(e->model[RLCF_REWARD + m].cfg);
    const bool overlap = sparse_ok && vit_rewards && !no_overlap && !e->no_side && !g_prof.enabled && e->side && prec_x3(e) && !prec_single(e);
    bool fwd_done = false;
    for (int j = 0; j < a->tta_steps; ++j) {
        // step 0 runs on ctx == ctx_init: its text features are the cached txt0 (the dense-backward
        // path still needs this pass for its saved activations)
        const bool cached = (j == 0 && sparse_ok && !a->ctx_in);
        if (!cached) TRY(text_forward(e, s, e->lay[0], e->tt, ctx, io, !sparse_ok, st));
        const float* txt_j = cached ? e->txt0.as<float>() : e->txt.as<float>();
        const float* rows_logits;
        if (j == 0) {   // tpt_cls_rl.py:57-59
            TRY(engine_logits(e, e->img_feat.as<float>(), N, txt_j, C, e->logits.as<float>(), st));
            TRY(launch_entropy_select(e->logits.as<float>(), N, C, n_sel, e->entropy.as<float>(), e->sel_idx.as<int32_t>(), st));
            TRY(launch_gather_rows(e->img_feat.as<float>(), D, e->sel_idx.as<int32_t>(), e->sel_feat.as<float>(), D, n_sel, D, st));
            TRY(launch_gather_rows(views, (int)img_elems, e->sel_idx.as<int32_t>(), e->views_sel.as<float>(), (int)img_elems, n_sel, (int)img_elems, st));
            // the reward models' pass over the selected views depends on nothing the student does from here to the loss, and at one
            // image's sizes neither it nor the sparse text forward fills the chip: second stream, joined before the loss kernel
            if (overlap) {
                // the side stream's own A-operand buffer: the patch matrix of the selected views or a <= 512-row token matrix against a
                // W x 4W weight, whichever reward model needs more (the main stream keeps a_hi for the text passes it runs meanwhile)
                size_t need2 = 0;
                for (int m = 0; m < e->n_rewards; ++m) {
                    const ClipModel& rm = e->model[RLCF_REWARD + m];
                    need2 = std::max(need2, (size_t)n_sel * rm.tokens * std::max(rm.Kp, 4 * rm.cfg.vision_width));
                }
                if (need2 > e->a_split2_elems) { TRY(e->a_hi2.ensure(need2 * 4)); e->a_split2_elems = need2; }
                RLCF_HIP_CHECK(hipEventRecord(e->ev_fork, st));
                RLCF_HIP_CHECK(hipStreamWaitEvent(e->side, e->ev_fork, 0));
                e->ws_sel = 1;
                const int rc_side = reward_encode(e, n_sel, s.cfg.image_resolution, out->reward_image_features, e->side);
                e->ws_sel = 0;
                const hipError_t er = hipEventRecord(e->ev_join, e->side);      // (recorded even after an error: the main stream must not run ahead)
                if (rc_side != RLCF_OK || er != hipSuccess) {
                    (void)hipStreamWaitEvent(st, e->ev_join, 0);
                    if (er != hipSuccess) { (void)hipStreamSynchronize(e->side); rlcf_set_error("hipEventRecord(ev_join): %s", hipGetErrorString(er)); return RLCF_ERR_HIP; }
                    return rc_side;
                }
            } else {
                TRY(reward_encode(e, n_sel, s.cfg.image_resolution, out->reward_image_features, st));
            }
            TRY(launch_gather_rows(e->logits.as<float>(), C, e->sel_idx.as<int32_t>(), e->sel_logits.as<float>(), C, n_sel, C, st));
            rows_logits = e->sel_logits.as<float>();
            if (overlap) {
                // whatever happens on the main stream, it joins the side stream before this call returns: the side stream writes
                // e->vt, e->rimg and the caller's reward_image_features
                int rc_main = launch_topk_rows(rows_logits, C, n_sel, C, K, e->topk_idx.as<int32_t>(), e->rl_stats.as<float>(), st);
                if (rc_main == RLCF_OK) rc_main = sparse_forward(e, ctx, e->topk_idx.as<int32_t>(), n_e, st);
                fwd_done = true;
                const hipError_t ej = hipStreamWaitEvent(st, e->ev_join, 0);
                if (rc_main != RLCF_OK) { if (ej != hipSuccess) (void)hipStreamSynchronize(e->side); return rc_main; }
                if (ej != hipSuccess) { (void)hipStreamSynchronize(e->side); rlcf_set_error("hipStreamWaitEvent(ev_join): %s", hipGetErrorString(ej)); return RLCF_ERR_HIP; }
            }
            COPY_OUT(out->logits, e->logits.p, (size_t)N * C * sizeof(float));
            COPY_OUT(out->entropy, e->entropy.p, N * sizeof(float));
            COPY_OUT(out->selected_idx, e->sel_idx.p, n_sel * sizeof(int32_t));
        } else {        // tpt_cls_rl.py:55 — selected views only; their image features are unchanged
            TRY(engine_logits(e, e->sel_feat.as<float>(), n_sel, txt_j, C, e->sel_logits.as<float>(), st));
            rows_logits = e->sel_logits.as<float>();
        }
        TRY(launch_reward_loss_bank(rows_logits, C, nullptr, 1, n_sel, C, K, reward_bank(e),
                               a->clipscore_weight, a->flags, a->min_entropy_w, e->topk_idx.as<int32_t>(), e->clip_score.as<float>(),
                               e->rewards.as<float>(), e->loss.as<float>(), e->dlogits.as<float>(), e->rl_stats.as<float>(), st));
        if (sparse_ok && fwd_done) {
            TRY(sparse_backward_only(e, e->sel_feat.as<float>(), e->topk_idx.as<int32_t>(), n_e, K, e->dlogits.as<float>(), e->ctx_grad.as<float>(), st));
            fwd_done = false;
        } else if (sparse_ok) {
            TRY(sparse_backward(e, ctx, e->sel_feat.as<float>(), e->topk_idx.as<int32_t>(), n_e, K, e->dlogits.as<float>(),
                                e->ctx_grad.as<float>(), st));
        } else {
            TRY(launch_dtxt_dense(e->dlogits.as<float>(), e->sel_feat.as<float>(), n_sel, C, D, s.logit_scale_exp, e->dtxt_dense.as<float>(), st));
            TRY(text_backward(e, s, e->tt, io, e->lay[0].max_keys, e->dtxt_dense.as<float>(), e->sp_du.as<float>(), e->sp_dxe.as<float>(),
                              e->lay[0].ctx_rows_list.as<int32_t>(), e->lay[0].n_copies, n_ctx, e->ctx_grad.as<float>(), st));
        }
        if (j == 0) {
            COPY_OUT(out->topk_idx, e->topk_idx.p, (size_t)n_e * sizeof(int32_t));
            COPY_OUT(out->clip_score, e->clip_score.p, (size_t)n_e * sizeof(float));
            COPY_OUT(out->rewards, e->rewards.p, (size_t)n_e * sizeof(float));
            COPY_OUT(out->loss, e->loss.p, sizeof(float));
            COPY_OUT(out->dlogits, e->dlogits.p, (size_t)n_sel * C * sizeof(float));
            COPY_OUT(out->ctx_grad, e->ctx_grad.p, cb);
        }
        // scaler.step(optimizer) (tpt_cls_rl.py:78): a gradient with an inf / NaN skips the update (the same inputs give the same
        // gradient at the following steps, so the host-side step number j + 1 never meets an applied step after a skipped one)
        TRY(launch_grad_nonfinite(e->ctx_grad.as<float>(), (int64_t)n_ctx * Wt, 1, e->step_skip.as<int32_t>(), st));
        TRY(launch_adamw(ctx, e->ctx_grad.as<float>(), e->adam_m.as<float>(), e->adam_v.as<float>(), (int64_t)n_ctx * Wt, j + 1, a->lr,
                         a->beta1, a->beta2, a->eps, a->weight_decay, st, e->step_skip.as<int32_t>(), (int64_t)n_ctx * Wt));
        if (out->step_skipped) COPY_OUT(out->step_skipped + j, e->step_skip.p, sizeof(int32_t));
    }
    // final inference on the clean view (views[0]) with the adapted prompt, tpt_cls_rl.py:260-262;
    // its image feature is row 0 of img_feat (frozen image tower: identical to re-encoding it).
    COPY_OUT(out->ctx_after, ctx, cb);
    if (a->skip_final) return RLCF_OK;
    TRY(text_forward(e, s, e->lay[0], e->tt, ctx, io, false, st));
    TRY(engine_logits(e, e->img_feat.as<float>(), 1, e->txt.as<float>(), C, e->final_logits.as<float>(), st));
    TRY(launch_top5(e->final_logits.as<float>(), C, e->top5.as<int32_t>(), st));
    COPY_OUT(out->final_logits, e->final_logits.p, (size_t)C * sizeof(float));
    COPY_OUT(out->top5, e->top5.p, 5 * sizeof(int32_t));
    return RLCF_OK;
}

// ------------------------------------------------------------------ B test samples per pass
// Same arithmetic per sample as engine_tta_sample (default RLCF configuration: one tuning step, sparse class backward),
// but every tower pass runs once for the whole batch: B*N views through the student image tower, B*n_sel views through
// the reward tower, B*n_sel*K class prompts through the sparse forward/backward (each sample with its own copy of the
// prompt prefix), B adapted prompts through one replicated final text pass.  Samples stay independent (no cross-sample
// arithmetic); larger M per GEMM is what fills 256 CUs.
static int batch_ensure(rlcf_engine* e, int B, hipStream_t st) {
    if (B <= e->b_cap) return RLCF_OK;
    ClipModel& s = e->model[RLCF_STUDENT];
    const TextLayout& L = e->lay[0];
    const int Wt = s.cfg.text_width, D = s.cfg.embed_dim, C = L.C;
    TRY(e->b_seqs_rep.ensure((size_t)B * L.n_seq * sizeof(rlcf_seq))); TRY(e->b_eot_rep.ensure((size_t)B * C * sizeof(int32_t)));
    TRY(launch_replicate_layout(L.seqs.as<rlcf_seq>(), L.n_seq, L.eot_rows.as<int32_t>(), C, L.T, B, e->b_seqs_rep.as<rlcf_seq>(),
                                e->b_eot_rep.as<int32_t>(), st));
    if (L.n_pk > 0) {        // packed runs: the descriptors shift like the sequences, the per-row sequence starts like row ids
        TRY(e->b_pk_rep.ensure((size_t)B * L.n_pk * sizeof(rlcf_seq))); TRY(e->b_rss_rep.ensure((size_t)B * L.T * sizeof(int32_t)));
        TRY(launch_replicate_layout(L.pk_seqs.as<rlcf_seq>(), L.n_pk, L.pk_rss.as<int32_t>(), L.T, L.T, B, e->b_pk_rep.as<rlcf_seq>(),
                                    e->b_rss_rep.as<int32_t>(), st));
    }
    const size_t cb = (size_t)B * e->n_ctx * Wt * sizeof(float);
    TRY(e->b_ctx.ensure(cb)); TRY(e->b_m.ensure(cb)); TRY(e->b_v.ensure(cb)); TRY(e->b_grad.ensure(cb));
    TRY(e->b_txt.ensure((size_t)B * C * D * sizeof(float))); TRY(e->b_u.ensure((size_t)B * C * D * sizeof(float)));
    TRY(e->b_eot_x.ensure((size_t)B * C * Wt * sizeof(float))); TRY(e->b_eot_ln.ensure((size_t)B * C * Wt * sizeof(float)));
    TRY(e->b_inv.ensure((size_t)B * C * sizeof(float))); TRY(e->b_logits.ensure((size_t)B * C * sizeof(float)));
    TRY(tower_ensure(e->tt, B * L.T, Wt, st));
    if (prec_x3(e) && (size_t)B * L.T * Wt * 4 > e->a_split_elems) {
        e->a_split_elems = (size_t)B * L.T * Wt * 4;
        TRY(e->a_hi.ensure(e->a_split_elems * 4));
    }
    RLCF_HIP_CHECK(hipStreamSynchronize(st));
    e->b_cap = B;
    return RLCF_OK;
}

// img_feat: the student image features [B*N, D] of these views — nullptr: computed here (the whole pass on one stream); else they
// were produced by the caller (tta_batch_pipelined: the tower of this part ran on the other stream) and only the rest of the pass runs
static int tta_batch_fused(rlcf_engine* e, const float* views, int B, int N, const rlcf_tta_args* a, float* final_logits, int32_t* top5,
                           hipStream_t st, const float* img_feat = nullptr) {
    ClipModel& s = e->model[RLCF_STUDENT];
    const TextLayout& L = e->lay[0];
    const int C = e->C, K = a->sample_k, D = s.cfg.embed_dim, Wt = s.cfg.text_width, n_ctx = e->n_ctx;
    const int n_sel = n_selected(a, N), n_e = n_sel * K, BN = B * N, BS = B * n_sel;
    const size_t img_elems = (size_t)3 * s.cfg.image_resolution * s.cfg.image_resolution;
    if (!img_feat) {
        TRY(batch_ensure(e, B, st));
        TRY(sparse_ensure(e, n_e, st, B));
        e->last_flops = 0.0;
        // 1. student image features of all B*N views
        TRY(engine_encode_image(e, RLCF_STUDENT, views, BN, e->img_feat.as<float>(), st));
        img_feat = e->img_feat.as<float>();
    }
    // first-step logits against the cached pristine-prompt text features
    TRY(engine_logits(e, img_feat, BN, e->txt0.as<float>(), C, e->logits.as<float>(), st));
    // 2. per-sample confidence selection (global row ids), gathers, reward features of the selected views
    TRY(launch_entropy_select_batched(e->logits.as<float>(), B, N, C, n_sel, e->entropy.as<float>(), e->sel_idx.as<int32_t>(), st));
    TRY(launch_gather_rows(img_feat, D, e->sel_idx.as<int32_t>(), e->sel_feat.as<float>(), D, BS, D, st));
    TRY(launch_gather_rows(views, (int)img_elems, e->sel_idx.as<int32_t>(), e->views_sel.as<float>(), (int)img_elems, BS, (int)img_elems, st));
    TRY(reward_encode(e, BS, s.cfg.image_resolution, nullptr, st));
    TRY(launch_gather_rows(e->logits.as<float>(), C, e->sel_idx.as<int32_t>(), e->sel_logits.as<float>(), C, BS, C, st));
    const int gT = L.pre_rows + n_e * L.lmax, T = B * gT, nE = B * n_e;
    const int64_t np = (int64_t)n_ctx * Wt;
    // reset state of every sample: ctx = ctx_init, Adam moments zero (custom_clip.py:161-164, tpt_cls_rl.py:251-255)
    TRY(launch_broadcast_rows(e->ctx_init.as<float>(), e->b_ctx.as<float>(), (int)np, B, st));
    RLCF_HIP_CHECK(hipMemsetAsync(e->b_m.p, 0, (size_t)B * np * sizeof(float), st));
    RLCF_HIP_CHECK(hipMemsetAsync(e->b_v.p, 0, (size_t)B * np * sizeof(float), st));
    TextPassIO fo{};                 // full class bank, one replica (and one prompt) per sample
    fo.seqs = e->b_seqs_rep.as<rlcf_seq>(); fo.n_seq = B * L.n_seq; fo.max_q_len = L.max_q_len; fo.T = B * L.T; fo.n_cls = B * C;
    fo.attn_pairs = (long)B * L.attn_pairs; fo.eot_rows = e->b_eot_rep.as<int32_t>(); fo.row_src = nullptr;
    fo.eot_x = e->b_eot_x.as<float>(); fo.eot_ln = e->b_eot_ln.as<float>(); fo.u = e->b_u.as<float>(); fo.inv_norm = e->b_inv.as<float>();
    fo.txt = e->b_txt.as<float>(); fo.rep_rows = L.T; fo.ctx_stride = (int)np;
    if (L.n_pk > 0) { fo.pk_seqs = e->b_pk_rep.as<rlcf_seq>(); fo.n_pk = B * L.n_pk; fo.pk_rss = e->b_rss_rep.as<int32_t>(); }
    for (int j = 0; j < a->tta_steps; ++j) {
        if (j > 0) {
            // tpt_cls_rl.py:55: logits of the selected views under each sample's current prompt
            TRY(text_forward(e, s, L, e->tt, e->b_ctx.as<float>(), fo, false, st));
            TRY(launch_group_logits(e->sel_feat.as<float>(), n_sel, e->b_txt.as<float>(), B, C, D, s.logit_scale_exp, e->sel_logits.as<float>(), st));
            e->last_flops += 2.0 * BS * C * D;
        }
        // 3. top-K sampling, CLIP reward, baseline, reward-weighted CE and dlogits, grouped per sample
        TRY(launch_reward_loss_bank(e->sel_logits.as<float>(), C, nullptr, B, n_sel, C, K, reward_bank(e),
                                    a->clipscore_weight, a->flags, a->min_entropy_w, e->topk_idx.as<int32_t>(), nullptr, nullptr, nullptr,
                                    e->dlogits.as<float>(), e->rl_stats.as<float>(), st));
        // 4. sparse backward of all B*n_e sampled (view, class) pairs; each sample owns a copy of the prompt prefix
        TRY(launch_build_sparse_layout(e->topk_idx.as<int32_t>(), B, n_e, L.class_start.as<int32_t>(), L.class_len.as<int32_t>(),
                                       L.class_eot_off.as<int32_t>(), L.lmax, L.pre_rows, e->sp_seqs.as<rlcf_seq>(), e->sp_eot_rows.as<int32_t>(),
                                       e->sp_row_src.as<int32_t>(), st));
        TextPassIO io{};
        io.seqs = e->sp_seqs.as<rlcf_seq>(); io.n_seq = B * (n_e + (L.pre_rows > 0 ? 1 : 0)); io.max_q_len = L.max_q_len; io.T = T; io.n_cls = nE;
        io.attn_pairs = (long)(nE * (L.mean_len * (L.pre_rows + (L.mean_len + 1) * 0.5)));
        io.eot_rows = e->sp_eot_rows.as<int32_t>(); io.row_src = e->sp_row_src.as<int32_t>();
        io.eot_x = e->sp_eot_x.as<float>(); io.eot_ln = e->sp_eot_ln.as<float>(); io.u = e->sp_u.as<float>();
        io.inv_norm = e->sp_inv_norm.as<float>(); io.txt = e->sp_txt.as<float>();
        io.rep_rows = gT; io.ctx_stride = (int)np;                  // group b of gT rows reads prompt b
        TRY(text_forward(e, s, L, e->st, e->b_ctx.as<float>(), io, true, st));
        TRY(launch_dtxt_sparse(e->dlogits.as<float>(), e->topk_idx.as<int32_t>(), e->sel_feat.as<float>(), nE, K, C, D, s.logit_scale_exp,
                               e->sp_dtxt.as<float>(), st));
        {   // text_backward with the per-sample (grouped) ctx-gradient reduction
            TRY(launch_l2norm_bwd(io.txt, e->sp_dtxt.as<float>(), io.inv_norm, e->sp_du.as<float>(), nE, D, st));
            TRY(gemm(e, e->sp_du.as<float>(), D, s.tproj, D, nullptr, nullptr, 0, nullptr, 0, e->sp_dxe.as<float>(), Wt, nE, Wt, D, 1.f, RLCF_EPI_NONE, st));
            TRY(launch_layernorm_bwd(io.eot_x, s.lnf_w, e->sp_dxe.as<float>(), nullptr, e->sp_dxe.as<float>(), nullptr, nullptr, nE, Wt, st));
            RLCF_HIP_CHECK(hipMemsetAsync(e->dX.p, 0, (size_t)T * Wt * sizeof(float), st));
            TRY(launch_scatter_rows(e->sp_dxe.as<float>(), io.eot_rows, e->dX.as<float>(), nE, Wt, st));
            TRY(transformer_backward(e, s.txt, e->st, io.seqs, io.n_seq, L.max_keys, io.attn_pairs, 1, T, st));
            if (L.ctx_general)
                TRY(launch_ctx_grad_scan(e->dX.as<float>(), io.row_src, L.ctx_row.as<int32_t>(), B, gT, n_ctx, Wt, e->b_grad.as<float>(), st));
            else
                TRY(launch_ctx_grad_grouped(e->dX.as<float>(), e->sp_ctx_rows_list.as<int32_t>(), L.pre_rows > 0 ? 1 : n_e, n_ctx, Wt, B, gT,
                                            e->b_grad.as<float>(), st));
        }
        // 5. AdamW step j+1 of every sample (tpt_cls_rl.py:76-79)
        TRY(launch_grad_nonfinite(e->b_grad.as<float>(), np, B, e->step_skip.as<int32_t>(), st));
        TRY(launch_adamw(e->b_ctx.as<float>(), e->b_grad.as<float>(), e->b_m.as<float>(), e->b_v.as<float>(), B * np, j + 1, a->lr, a->beta1,
                         a->beta2, a->eps, a->weight_decay, st, e->step_skip.as<int32_t>(), np));
    }
    // 6. final clean-view inference: B adapted prompts through one replicated text pass
    TRY(text_forward(e, s, L, e->tt, e->b_ctx.as<float>(), fo, false, st));
    float* fl = final_logits ? final_logits : e->b_logits.as<float>();
    TRY(launch_final_logits_batched(img_feat, N, e->b_txt.as<float>(), B, C, D, s.logit_scale_exp, fl, st));
    e->last_flops += 2.0 * B * C * D;
    TRY(launch_top5_batched(fl, B, C, top5, st));
    return RLCF_OK;
}

// One pass of B test images in `parts` parts on TWO streams: the student image tower of part k+1 (chip-filling GEMMs) runs on the
// caller's stream while everything behind the tower of part k — reward models' pass over the selected views, loss, sparse text
// forward / backward, AdamW, the replicated final text pass: small launch-bound kernels, ~20 % of a pass — runs on the side stream
// with the side stream's own scratch (ws_sel: A-operand buffer, split-K workspace, image-tower scratch).  Samples are independent, so
// the per-sample results are those of the one-stream pass up to the round-off of the GEMM forms the part sizes select
// (test_batch_pipeline_equals_single_stream).  MEASURED SLOWER than the one-stream pass on BASELINE configs[1] (115.7 images/s
// against 114.1 / 109.3 in 2 / 4 parts: a workgroup of the small kernels blocks a CU for the 139-KB GEMM workgroups of the tower just as
// it does alone, so the two-stream form hides nothing) — built only when RLCF_BATCH_PARTS=n asks for it.
static int tta_batch_pipelined(rlcf_engine* e, const float* views, int B, int N, int parts, const rlcf_tta_args* a, float* final_logits,
                               int32_t* top5, hipStream_t st) {
    ClipModel& s = e->model[RLCF_STUDENT];
    const int D = s.cfg.embed_dim, n_sel = n_selected(a, N), n_e = n_sel * a->sample_k;
    const size_t per = (size_t)N * 3 * s.cfg.image_resolution * s.cfg.image_resolution;
    const int Bp = (B + parts - 1) / parts;
    TRY(batch_ensure(e, Bp, st));
    TRY(sparse_ensure(e, n_e, st, Bp));
    {   // side-stream operand buffer: the text passes of a part and the reward models' patch matrices
        size_t need2 = (size_t)Bp * e->lay[0].T * s.cfg.text_width * 4;
        for (int m = 0; m < e->n_rewards; ++m) {
            const ClipModel& rm = e->model[RLCF_REWARD + m];
            need2 = std::max(need2, (size_t)Bp * n_sel * rm.tokens * std::max(rm.Kp, 4 * rm.cfg.vision_width));
        }
        if (need2 > e->a_split2_elems) { TRY(e->a_hi2.ensure(need2 * 4)); e->a_split2_elems = need2; }
    }
    for (int k = 0; k < parts && k < 8; ++k)
        if (!e->ev_part[k]) RLCF_HIP_CHECK(hipEventCreateWithFlags(&e->ev_part[k], hipEventDisableTiming));
    e->last_flops = 0.0;
    RLCF_HIP_CHECK(hipEventRecord(e->ev_fork, st));
    RLCF_HIP_CHECK(hipStreamWaitEvent(e->side, e->ev_fork, 0));
    int rc = RLCF_OK;
    for (int k = 0, b0 = 0; k < parts && b0 < B && rc == RLCF_OK; ++k, b0 += Bp) {
        const int Bk = std::min(Bp, B - b0);
        float* feat_k = e->img_feat.as<float>() + (size_t)b0 * N * D;
        rc = engine_encode_image(e, RLCF_STUDENT, views + (size_t)b0 * per, Bk * N, feat_k, st);
        if (rc != RLCF_OK) break;
        if (hipEventRecord(e->ev_part[k], st) != hipSuccess || hipStreamWaitEvent(e->side, e->ev_part[k], 0) != hipSuccess) { rc = RLCF_ERR_HIP; break; }
        e->ws_sel = 1;
        rc = tta_batch_fused(e, views + (size_t)b0 * per, Bk, N, a, final_logits ? final_logits + (size_t)b0 * e->C : nullptr, top5 + (size_t)b0 * 5,
                             e->side, feat_k);
        e->ws_sel = 0;
    }
    // the caller's stream joins the side stream whatever happened
    const hipError_t e1 = hipEventRecord(e->ev_join, e->side);
    const hipError_t e2 = e1 == hipSuccess ? hipStreamWaitEvent(st, e->ev_join, 0) : e1;
    if (e2 != hipSuccess) { (void)hipStreamSynchronize(e->side); if (rc == RLCF_OK) { rlcf_set_error("tta_batch_pipelined: join: %s", hipGetErrorString(e2)); rc = RLCF_ERR_HIP; } }
    return rc;
}

int engine_tta_batch(rlcf_engine* e, const float* views, int count, int N, const rlcf_tta_args* a, float* final_logits, int32_t* top5,
                     hipStream_t st) {
    ClipModel& s = e->model[RLCF_STUDENT];
    if (e->C <= 0 || e->image_bank || e->n_rewards <= 0) { rlcf_set_error("class bank / reward model not set"); return RLCF_ERR_STATE; }
    if (e->n_ctx <= 0) { rlcf_set_error("prompt tuning needs a class bank with learnable context rows (n_ctx > 0)"); return RLCF_ERR_STATE; }
    RLCF_ARG_CHECK(N > 0 && N <= e->max_views && a->sample_k > 0 && a->sample_k <= 32 && a->sample_k <= e->C);
    const size_t per = (size_t)N * 3 * s.cfg.image_resolution * s.cfg.image_resolution;
    const int n_sel = n_selected(a, N);
    const bool sparse_ok = a->sparse_backward && (a->flags & RLCF_F_REWARD_PROCESS) && !(a->flags & RLCF_F_PROCESS_BATCH) &&
                           !(a->flags & RLCF_F_MIN_ENTROPY) && a->sample_k > 1;
    const int Bmax = e->max_views / N;
    const bool fused = Bmax >= 2 && a->tta_steps >= 1 && sparse_ok && !a->ctx_in && !a->skip_final && n_sel > 0;
    double flops = 0.0;
    int i = 0;
    while (i < count) {
        const int B = fused ? std::min(Bmax, count - i) : 1;
        if (fused && B >= 2) {
            // parts of a pass on two streams (tta_batch_pipelined): needs the side stream, ViT towers everywhere (the ModifiedResNet
            // pass keeps its scratch in the engine), no per-launch profile (its event pairs serialise), one tuning step
            static int parts_env = -1;
            if (parts_env < 0) { const char* ev = getenv("RLCF_BATCH_PARTS"); parts_env = ev ? atoi(ev) : 0; }
            bool vit_all = !is_resnet(s.cfg);
            for (int m = 0; m < e->n_rewards; ++m) vit_all = vit_all && !is_resnet(e->model[RLCF_REWARD + m].cfg);
            int parts = parts_env > 0 ? parts_env : 1;          // (measured slower than one stream on BASELINE configs[1]: off unless asked for)
            parts = std::min(std::min(parts, 8), B / 2);
            if (parts >= 2 && vit_all && e->side && !g_prof.enabled && prec_x3(e)) {
                TRY(tta_batch_pipelined(e, views + (size_t)i * per, B, N, parts, a, final_logits ? final_logits + (size_t)i * e->C : nullptr,
                                        top5 + (size_t)i * 5, st));
            } else
            TRY(tta_batch_fused(e, views + (size_t)i * per, B, N, a, final_logits ? final_logits + (size_t)i * e->C : nullptr, top5 + (size_t)i * 5, st));
        } else {
            rlcf_tta_out o{};
            o.top5 = top5 + (size_t)i * 5;
            o.final_logits = final_logits ? final_logits + (size_t)i * e->C : nullptr;
            TRY(engine_tta_sample(e, views + (size_t)i * per, N, a, &o, st));
        }
        flops += e->last_flops;
        i += B;
    }
    e->last_flops = flops / count;
    return RLCF_OK;
}
