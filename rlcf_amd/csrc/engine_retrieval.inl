// Part of engine.hip (one translation unit: #include'd there): text-encoder tuning of the retrieval policy, text -> image
// (retrieval/clip_ret_policy.py:106-137,193-196 with CLIPRet_TTA(only_visual=False)).
// ------------------------------------------------------------------ text-encoder tuning (retrieval, text -> image)
// retrieval/clip_ret_policy.py:106-137 (tune_text) with CLIPRet_TTA(only_visual=False): parameters() = every parameter of the CLIP
// model whose name does not contain 'visual' (custom_models.py:139-147) — token_embedding.weight, positional_embedding, the text
// transformer, ln_final, text_projection and logit_scale — tuned on ONE query caption against a fixed bank of image features.
// Same buffer scheme as the image encoder (engine_visual_enable): the LayerNorm tensors in one small buffer, everything else in a
// flat one, the tower reads the live weights from them, derived copies (transposes, split-f16 pairs, the query's embedding rows)
// follow after every optimizer step and after the reset.
int engine_text_enable(rlcf_engine* e, hipStream_t st) {
    if (e->tw_count) return RLCF_OK;
    ClipModel& m = e->model[RLCF_STUDENT];
    if (!m.finalized) { rlcf_set_error("student model not finalized"); return RLCF_ERR_STATE; }
    if (prec_single(e)) { rlcf_set_error("encoder tuning runs in RLCF_PREC_F32 / RLCF_PREC_F16X3 (RLCF_PREC_F16 is the prompt path's performance mode)"); return RLCF_ERR_STATE; }
    const rlcf_clip_cfg& c = m.cfg;
    const size_t Wt = c.text_width, D = c.embed_dim, W2 = Wt * Wt;
    struct Item { const float** slot; size_t numel; };
    std::vector<Item> items = {{&m.tok_emb, (size_t)c.vocab_size * Wt}, {&m.tpos, (size_t)c.context_length * Wt}, {&m.tproj, Wt * D}};
    for (BlockW& b : m.txt.blk) {
        items.push_back({&b.in_w, 3 * W2}); items.push_back({&b.in_b, 3 * Wt}); items.push_back({&b.out_w, W2}); items.push_back({&b.out_b, Wt});
        items.push_back({&b.fc_w, 4 * W2}); items.push_back({&b.fc_b, 4 * Wt}); items.push_back({&b.proj_w, 4 * W2}); items.push_back({&b.proj_b, Wt});
    }
    size_t total = 0;
    e->tw_slots.clear();
    for (const Item& it : items) { e->tw_slots.push_back(VwSlot{total, it.numel}); total += (it.numel + 63) / 64 * 64; }
    e->tw_slots.push_back(VwSlot{total, 1});              // logit_scale (the parameter, not its exponential)
    total += 64;
    const size_t nb = total * sizeof(float);
    for (DevBuf* d : {&e->tw, &e->tw_init, &e->tw_grad, &e->tw_m, &e->tw_v}) TRY(d->ensure(nb));
    RLCF_HIP_CHECK(hipMemsetAsync(e->tw.p, 0, nb, st));
    for (size_t i = 0; i < items.size(); ++i) {
        const float* old = *items[i].slot;
        float* dst = e->tw.as<float>() + e->tw_slots[i].off;
        RLCF_HIP_CHECK(hipMemcpyAsync(dst, old, items[i].numel * sizeof(float), hipMemcpyDeviceToDevice, st));
        auto sp = m.split_of.find(old);
        if (sp != m.split_of.end()) { const ClipModel::SplitW s = sp->second; m.split_of.erase(sp); m.split_of[dst] = s; }
        *items[i].slot = dst;
    }
    const float* ls = rawp(m, "logit_scale", 1);
    NEED(ls);
    RLCF_HIP_CHECK(hipMemcpyAsync(e->tw.as<float>() + e->tw_slots.back().off, ls, sizeof(float), hipMemcpyDeviceToDevice, st));
    RLCF_HIP_CHECK(hipMemcpyAsync(e->tw_init.p, e->tw.p, nb, hipMemcpyDeviceToDevice, st));
    for (DevBuf* d : {&e->tw_clip, &e->tw_mom}) { TRY(d->ensure(nb)); RLCF_HIP_CHECK(hipMemcpyAsync(d->p, e->tw.p, nb, hipMemcpyDeviceToDevice, st)); }
    // LayerNorms: [ln_final.weight | ln_final.bias | per block ln_1.weight ln_1.bias ln_2.weight ln_2.bias] (transformer_backward's layout)
    const int L = c.text_layers;
    e->tln_count = (int)((4 * L + 2) * Wt);
    const size_t lb = (size_t)e->tln_count * sizeof(float);
    for (DevBuf* d : {&e->tln, &e->tln_init, &e->tln_grad, &e->tln_m, &e->tln_v}) TRY(d->ensure(lb));
    {
        float* P = e->tln.as<float>();
        std::vector<const float**> slots = {&m.lnf_w, &m.lnf_b};
        for (BlockW& b : m.txt.blk) { slots.push_back(&b.ln1_w); slots.push_back(&b.ln1_b); slots.push_back(&b.ln2_w); slots.push_back(&b.ln2_b); }
        for (size_t i = 0; i < slots.size(); ++i) {
            RLCF_HIP_CHECK(hipMemcpyAsync(P + i * Wt, *slots[i], Wt * sizeof(float), hipMemcpyDeviceToDevice, st));
            *slots[i] = P + i * Wt;
        }
        RLCF_HIP_CHECK(hipMemcpyAsync(e->tln_init.p, P, lb, hipMemcpyDeviceToDevice, st));
        for (DevBuf* d : {&e->tln_clip, &e->tln_mom}) { TRY(d->ensure(lb)); RLCF_HIP_CHECK(hipMemcpyAsync(d->p, P, lb, hipMemcpyDeviceToDevice, st)); }
    }
    e->tw_refresh.clear();
    auto add_split = [&](const float* w, size_t numel) {
        auto it = m.split_of.find(w);
        if (it != m.split_of.end()) it->second.lo_zero = false;          // (a TUNED weight leaves the fp16 grid at its first step: three passes)
        if (it != m.split_of.end())
            e->tw_refresh.push_back(VwRefresh{VW_SPLIT, w, nullptr, numel, 0, it->second.hi, it->second.lo, 1.0f / it->second.inv_scale,
                                              it->second.lo == lo_of(it->second.hi)});
    };
    auto add_T = [&](const float* w, const float* wT, size_t rows, size_t cols) {
        if (!wT) return;
        e->tw_refresh.push_back(VwRefresh{VW_TRANSPOSE, w, (float*)wT, rows, cols, nullptr, nullptr, 1.f, 0});
        add_split(wT, rows * cols);
    };
    add_T(m.tproj, m.tprojT, Wt, D);
    for (BlockW& b : m.txt.blk) {
        add_split(b.in_w, 3 * W2); add_split(b.out_w, W2); add_split(b.fc_w, 4 * W2); add_split(b.proj_w, 4 * W2);
        add_T(b.in_w, b.in_wT, 3 * Wt, Wt); add_T(b.out_w, b.out_wT, Wt, Wt); add_T(b.fc_w, b.fc_wT, 4 * Wt, Wt); add_T(b.proj_w, b.proj_wT, Wt, 4 * Wt);
    }
    e->tw_count = total;
    e->tw_dirty = false;
    RLCF_HIP_CHECK(hipStreamSynchronize(st));
    return RLCF_OK;
}

// x[i] *= exp(*log_scale)
__global__ void scale_by_exp_kernel(float* __restrict__ x, const float* __restrict__ log_scale, int n) {
    const float s = expf(*log_scale);
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) x[i] *= s;
}
// out[0] = sum_i a[i] * b[i]  (one block, fixed reduction order)
__global__ void dot_kernel(const float* __restrict__ a, const float* __restrict__ b, int n, float* __restrict__ out) {
    __shared__ float red[256];
    float acc = 0.f;
    for (int i = threadIdx.x; i < n; i += 256) acc += a[i] * b[i];
    red[threadIdx.x] = acc;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
        __syncthreads();
    }
    if (threadIdx.x == 0) out[0] = red[0];
}
// embedding gradients of one packed text: x[r] = token_embedding[token[r]] + positional_embedding[pos[r]] (model.py:344-346), so
// d positional_embedding[pos[r]] += dX[r], d token_embedding[token[r]] += dX[r].  One thread per column walks the rows in order:
// repeated tokens accumulate in a fixed order, no atomics.
__global__ void embed_grad_kernel(const float* __restrict__ dX, const int32_t* __restrict__ row_token, const int32_t* __restrict__ row_pos,
                                  float* __restrict__ g_tok, float* __restrict__ g_pos, int rows, int width) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= width) return;
    for (int r = 0; r < rows; ++r) {
        const float g = dX[(size_t)r * width + c];
        g_pos[(size_t)row_pos[r] * width + c] += g;
        const int t = row_token[r];
        if (t >= 0) g_tok[(size_t)t * width + c] += g;
    }
}

// the bank of the text -> image direction: n images, their features under the student (the fixed side of logits_per_text,
// CLIPRet_TTA.set_image_features, custom_models.py:91-95) and under every reward model (CLIPRewards.set_image_features,
// retrieval/clip_reward.py:130-137), blocks [n, Dr_m] one after another.  Device pointers, copied.
int engine_set_image_bank(rlcf_engine* e, const float* student_feats, const float* reward_feats, int n, hipStream_t st) {
    ClipModel& s = e->model[RLCF_STUDENT];
    if (!s.finalized || e->n_rewards <= 0) { rlcf_set_error("student / reward model not set"); return RLCF_ERR_STATE; }
    RLCF_ARG_CHECK(student_feats && reward_feats && n > 0 && n <= e->max_classes);
    const int D = s.cfg.embed_dim;
    e->C = n; e->image_bank = true; e->n_ctx = 0;
    e->sp_max_e = 0; e->sp_groups = 0; e->b_cap = 0;
    TRY(tta_scratch_ensure(e, n));
    TRY(e->txt0.ensure((size_t)n * D * sizeof(float)));
    RLCF_HIP_CHECK(hipMemcpyAsync(e->txt0.p, student_feats, (size_t)n * D * sizeof(float), hipMemcpyDeviceToDevice, st));
    for (int m = 0; m < e->n_rewards; ++m) {
        ClipModel& r = e->model[RLCF_REWARD + m];
        if (!r.finalized) { rlcf_set_error("reward model %d not finalized", m); return RLCF_ERR_STATE; }
        const int Dr = r.cfg.embed_dim;
        TRY(e->rimg[m].ensure((size_t)e->max_views * Dr * sizeof(float)));
        TRY(e->reward_cls[m].ensure((size_t)n * Dr * sizeof(float)));
        RLCF_HIP_CHECK(hipMemcpyAsync(e->reward_cls[m].p, reward_feats, (size_t)n * Dr * sizeof(float), hipMemcpyDeviceToDevice, st));
        reward_feats += (size_t)n * Dr;
    }
    RLCF_HIP_CHECK(hipStreamSynchronize(st));
    return RLCF_OK;
}

int engine_text_reset(rlcf_engine* e, hipStream_t st, bool force) {      // clip_model.load_state_dict(initial_state_dict) for the text side
    if (!e->tw_count || (!e->tw_dirty && !force)) return RLCF_OK;
    RLCF_HIP_CHECK(hipMemcpyAsync(e->tw.p, e->tw_init.p, e->tw_count * sizeof(float), hipMemcpyDeviceToDevice, st));
    RLCF_HIP_CHECK(hipMemcpyAsync(e->tln.p, e->tln_init.p, (size_t)e->tln_count * sizeof(float), hipMemcpyDeviceToDevice, st));
    e->tw_dirty = false;
    return refresh_derived(e->tw_refresh, 0, st);
}
static int rebuild_E(ClipModel& m, TextLayout& L, hipStream_t st) {
    build_E_kernel<<<dim3(L.T), dim3(128), 0, st>>>(m.tok_emb, m.tpos, L.row_token.as<int32_t>(), L.row_pos.as<int32_t>(), L.E.as<float>(), L.T,
                                                     m.cfg.text_width);
    RLCF_LAUNCH_CHECK();
    return RLCF_OK;
}

// tokens: HOST [context_length], the query caption.  Per step: logits_per_text [1, n] of the query against the image bank, top-K
// images, CLIPScore(images_index=...) of the reward model, REINFORCE loss, backward through the whole text encoder, AdamW over the
// two parameter buffers.  Then logits_per_text of the tuned encoder (clip_ret_policy.py:193-195) and the reset (:196).
// Outputs (rlcf_tta_out): logits [n], topk_idx, clip_score, rewards, loss, dlogits [n] of the first step; vis_grad / vis_after = the flat
// text buffer (rlcf_engine_text_param_layout), ln_grad / ln_after = the text LayerNorm buffer; final_logits [n]; step_skipped.
int engine_tta_retrieval_text(rlcf_engine* e, const int32_t* tokens, const rlcf_tta_args* a, const rlcf_tta_out* out, hipStream_t st) {
    ClipModel& s = e->model[RLCF_STUDENT];
    if (e->C <= 0 || !e->image_bank || e->n_rewards <= 0) { rlcf_set_error("image bank / reward model not set"); return RLCF_ERR_STATE; }
    RLCF_ARG_CHECK(tokens && a && a->tta_steps >= 0 && a->sample_k > 0 && a->sample_k <= 32 && a->sample_k <= e->C);
    TRY(engine_text_enable(e, st));
    const int C = e->C, K = a->sample_k, D = s.cfg.embed_dim, Wt = s.cfg.text_width, L = s.cfg.text_layers;
    const rlcf_tta_out none{};
    if (!out) out = &none;
    e->last_flops = 0.0;
    // model.reset_initial() + a fresh optimizer state (clip_ret_policy.py:186-190)
    TRY(engine_text_reset(e, st, false));
    const size_t wb = e->tw_count * sizeof(float), lb = (size_t)e->tln_count * sizeof(float);
    for (DevBuf* d : {&e->tw_m, &e->tw_v}) RLCF_HIP_CHECK(hipMemsetAsync(d->p, 0, wb, st));
    for (DevBuf* d : {&e->tln_m, &e->tln_v}) RLCF_HIP_CHECK(hipMemsetAsync(d->p, 0, lb, st));
    // layouts of the query under the student and under every reward model (one packed sequence each)
    TextLayout& Q = e->qlay[0];
    TRY(build_layout(e, s, Q, tokens, 1, 0, false, RLCF_TEXT_PACKED, st));
    int Wmax = Wt, Dmax = D, Tmax = Q.T;
    for (int m = 0; m < e->n_rewards; ++m) {
        ClipModel& r = e->model[RLCF_REWARD + m];
        RLCF_ARG_CHECK(r.cfg.context_length == s.cfg.context_length);
        TRY(build_layout(e, r, e->qlay[1 + m], tokens, 1, 0, false, RLCF_TEXT_PACKED, st));
        Wmax = std::max(Wmax, r.cfg.text_width); Dmax = std::max(Dmax, r.cfg.embed_dim); Tmax = std::max(Tmax, e->qlay[1 + m].T);
    }
    TRY(tower_ensure(e->tt, Tmax, Wmax, st));
    TRY(tower_ensure(e->st, Q.T, Wt, st));
    TRY(tower_ensure_saved(e->st, Q.T, Wt, L, st));
    TRY(bwd_ensure(e, Q.T, Wt));
    if (prec_x3(e) && (size_t)Tmax * Wmax * 4 > e->a_split_elems) {
        e->a_split_elems = (size_t)Tmax * Wmax * 4;
        TRY(e->a_hi.ensure(e->a_split_elems * 4));
    }
    TRY(e->eot_x.ensure((size_t)Wmax * sizeof(float))); TRY(e->eot_ln.ensure((size_t)Wmax * sizeof(float)));
    TRY(e->u.ensure((size_t)Dmax * sizeof(float))); TRY(e->inv_norm.ensure(sizeof(float))); TRY(e->txt.ensure((size_t)Dmax * sizeof(float)));
    TRY(e->q_feat.ensure((size_t)D * sizeof(float))); TRY(e->q_dfeat.ensure((size_t)D * sizeof(float))); TRY(e->q_ls.ensure(64 * sizeof(float)));
    TRY(e->sp_du.ensure((size_t)D * sizeof(float))); TRY(e->sp_dxe.ensure((size_t)Wt * sizeof(float)));
    auto query_io = [&](const TextLayout& Lq, float* txt) {
        TextPassIO io{};
        io.seqs = Lq.seqs.as<rlcf_seq>(); io.n_seq = Lq.n_seq; io.max_q_len = Lq.max_q_len; io.T = Lq.T; io.n_cls = 1;
        io.attn_pairs = Lq.attn_pairs; io.eot_rows = Lq.eot_rows.as<int32_t>(); io.row_src = nullptr;
        io.eot_x = e->eot_x.as<float>(); io.eot_ln = e->eot_ln.as<float>(); io.u = e->u.as<float>(); io.inv_norm = e->inv_norm.as<float>();
        io.txt = txt;
        return io;
    };
    // reward_model.set_text_features(captions=text) (clip_ret_policy.py:117): the query under every reward model, frozen
    for (int m = 0; m < e->n_rewards; ++m)
        TRY(text_forward(e, e->model[RLCF_REWARD + m], e->qlay[1 + m], e->tt, nullptr, query_io(e->qlay[1 + m], e->rimg[m].as<float>()), false, st));
    const TextPassIO io = query_io(Q, e->q_feat.as<float>());
    float* const ls = e->tw.as<float>() + e->tw_slots.back().off;
    float* const G = e->tw_grad.as<float>();
    float* const GL = e->tln_grad.as<float>();
    auto logits_per_text = [&](float* dst) -> int {
        TRY(gemm(e, e->q_feat.as<float>(), D, e->txt0.as<float>(), D, nullptr, nullptr, 0, nullptr, 0, dst, C, 1, C, D, 1.f, RLCF_EPI_NONE, st));
        scale_by_exp_kernel<<<dim3(std::min((C + 255) / 256, 1024)), dim3(256), 0, st>>>(dst, ls, C);
        RLCF_LAUNCH_CHECK();
        return RLCF_OK;
    };
    for (int j = 0; j < a->tta_steps; ++j) {
        TRY(text_forward(e, s, Q, e->st, nullptr, io, true, st));
        TRY(logits_per_text(e->sel_logits.as<float>()));
        TRY(launch_reward_loss_bank(e->sel_logits.as<float>(), C, nullptr, 1, 1, C, K, reward_bank(e), a->clipscore_weight, a->flags,
                                    a->min_entropy_w, e->topk_idx.as<int32_t>(), e->clip_score.as<float>(), e->rewards.as<float>(),
                                    e->loss.as<float>(), e->dlogits.as<float>(), e->rl_stats.as<float>(), st));
        RLCF_HIP_CHECK(hipMemsetAsync(G, 0, wb, st));
        RLCF_HIP_CHECK(hipMemsetAsync(GL, 0, lb, st));
        // logits = exp(logit_scale) * <feat, bank>:  d logit_scale = <dlogits, logits>,  d feat = exp(logit_scale) * dlogits @ bank
        dot_kernel<<<dim3(1), dim3(256), 0, st>>>(e->dlogits.as<float>(), e->sel_logits.as<float>(), C, G + e->tw_slots.back().off);
        RLCF_LAUNCH_CHECK();
        TRY(launch_dimg(e->dlogits.as<float>(), e->txt0.as<float>(), 1, C, D, 1.0f, e->q_dfeat.as<float>(), st));
        scale_by_exp_kernel<<<dim3(1), dim3(256), 0, st>>>(e->q_dfeat.as<float>(), ls, D);
        RLCF_LAUNCH_CHECK();
        // text_features = normalize(ln_final(x[eot]) @ text_projection) (model.py:351-356, custom_models.py:82-83)
        float *du = e->sp_du.as<float>(), *dxe = e->sp_dxe.as<float>();
        TRY(launch_l2norm_bwd(io.txt, e->q_dfeat.as<float>(), io.inv_norm, du, 1, D, st));
        TRY(wgrad(e, io.eot_ln, Wt, Wt, du, D, D, 1, G + e->tw_slots[2].off, nullptr, st));
        TRY(gemm(e, du, D, s.tproj, D, nullptr, nullptr, 0, nullptr, 0, dxe, Wt, 1, Wt, D, 1.f, RLCF_EPI_NONE, st));
        TRY(launch_layernorm_bwd(io.eot_x, s.lnf_w, dxe, nullptr, dxe, GL, GL + Wt, 1, Wt, st));
        RLCF_HIP_CHECK(hipMemsetAsync(e->dX.p, 0, (size_t)io.T * Wt * sizeof(float), st));
        TRY(launch_scatter_rows(dxe, io.eot_rows, e->dX.as<float>(), 1, Wt, st));
        TRY(transformer_backward(e, s.txt, e->st, io.seqs, io.n_seq, Q.max_keys, io.attn_pairs, 1, io.T, st, GL, 0, 0, 0, G, e->tw_slots.data() + 3));
        embed_grad_kernel<<<dim3((Wt + 127) / 128), dim3(128), 0, st>>>(e->dX.as<float>(), Q.row_token.as<int32_t>(), Q.row_pos.as<int32_t>(),
                                                                       G + e->tw_slots[0].off, G + e->tw_slots[1].off, io.T, Wt);
        RLCF_LAUNCH_CHECK();
        if (j == 0) {
            COPY_OUT(out->logits, e->sel_logits.p, (size_t)C * sizeof(float));
            COPY_OUT(out->topk_idx, e->topk_idx.p, (size_t)K * sizeof(int32_t));
            COPY_OUT(out->clip_score, e->clip_score.p, (size_t)K * sizeof(float));
            COPY_OUT(out->rewards, e->rewards.p, (size_t)K * sizeof(float));
            COPY_OUT(out->loss, e->loss.p, sizeof(float));
            COPY_OUT(out->dlogits, e->dlogits.p, (size_t)C * sizeof(float));
            COPY_OUT(out->vis_grad, G, wb);
            COPY_OUT(out->ln_grad, GL, lb);
            COPY_OUT(out->reward_image_features, e->rimg[0].p, (size_t)e->model[RLCF_REWARD].cfg.embed_dim * sizeof(float));
        }
        TRY(launch_grad_nonfinite(GL, e->tln_count, 1, e->step_skip.as<int32_t>(), st));
        TRY(launch_grad_nonfinite(G, (int64_t)e->tw_count, 1, e->step_skip.as<int32_t>(), st, true));
        if (out->step_skipped) COPY_OUT(out->step_skipped + j, e->step_skip.p, sizeof(int32_t));
        TRY(launch_adamw(e->tln.as<float>(), GL, e->tln_m.as<float>(), e->tln_v.as<float>(), e->tln_count, j + 1, a->lr, a->beta1, a->beta2,
                         a->eps, a->weight_decay, st, e->step_skip.as<int32_t>(), e->tln_count));
        TRY(launch_adamw(e->tw.as<float>(), G, e->tw_m.as<float>(), e->tw_v.as<float>(), (int64_t)e->tw_count, j + 1, a->lr, a->beta1, a->beta2,
                         a->eps, a->weight_decay, st, e->step_skip.as<int32_t>(), (int64_t)e->tw_count));
        e->tw_dirty = true;
        TRY(refresh_derived(e->tw_refresh, 0, st));
        TRY(rebuild_E(s, Q, st));
    }
    COPY_OUT(out->vis_after, e->tw.p, wb);
    COPY_OUT(out->ln_after, e->tln.p, lb);
    if (!a->skip_final) {
        TRY(text_forward(e, s, Q, e->st, nullptr, io, false, st));
        TRY(logits_per_text(e->final_logits.as<float>()));
        COPY_OUT(out->final_logits, e->final_logits.p, (size_t)C * sizeof(float));
    }
    TRY(engine_text_reset(e, st, false));
    return RLCF_OK;
}
