// Part of engine.hip (one translation unit: #include'd there): image-encoder tuning — the LayerNorm-tuning step (BASELINE configs[2],
// TPT/tune_cls_rl.py with CLIPCLS_TTA(only_norm=True)), BatchNorm tuning of a ModifiedResNet student, every-parameter tuning.
// ------------------------------------------------------------------ LayerNorm-tuning step (BASELINE configs[2])
// Image tower forward of n views WITH saved activations (CLIPCLS_TTA.forward, custom_clip.py:423-432).
static int vit_forward_saved(rlcf_engine* e, ClipModel& m, const float* images, int n, float* feats, hipStream_t st) {
    const rlcf_clip_cfg& c = m.cfg;
    const int Wv = c.vision_width, tok = m.tokens, G2 = tok - 1, T = n * tok, D = c.embed_dim;
    TRY(tower_ensure_saved(e->vt, T, Wv, c.vision_layers, st));
    TRY(launch_im2col(images, e->patches.as<float>(), nullptr, nullptr, n, c.image_resolution, c.vision_patch_size, m.Kp, st));
    TRY(gemm(e, e->patches.as<float>(), m.Kp, m.conv_w, m.Kp, nullptr, nullptr, 0, nullptr, 0, e->patch_out.as<float>(), Wv, n * G2, Wv, m.Kp,
             1.f, RLCF_EPI_NONE, st));
    {
        const LnRef gw = ln_ref(e, m.lnpre_w, 1), gb = ln_ref(e, m.lnpre_b, 1);
        TRY(launch_vit_assemble(e->patch_out.as<float>(), m.cls, m.vpos, gw.p, gb.p, e->vt.sv[0].x, n, tok, Wv, st, gw.group_rows, gw.group_stride));
    }
    TRY(transformer_forward(e, m.vis, e->vt, e->vit_seqs.as<rlcf_seq>(), n, tok, (long)n * tok * tok, 0, T, true, st));
    TRY(launch_gather_rows(e->vt.x.as<float>(), tok * Wv, nullptr, e->cls_rows.as<float>(), Wv, n, Wv, st));
    {
        const LnRef gw = ln_ref(e, m.lnpost_w, 1), gb = ln_ref(e, m.lnpost_b, 1);
        TRY(launch_layernorm_fwd(e->cls_rows.as<float>(), gw.p, gb.p, e->cls_ln.as<float>(), n, Wv, st, gw.group_rows, gw.group_stride));
    }
    TRY(gemm(e, e->cls_ln.as<float>(), Wv, m.vprojT, Wv, nullptr, nullptr, 0, nullptr, 0, e->feat_raw.as<float>(), D, n, D, Wv, 1.f,
             RLCF_EPI_NONE, st));
    TRY(launch_l2norm_rows(e->feat_raw.as<float>(), feats, e->vit_inv_norm.as<float>(), n, D, st));
    return RLCF_OK;
}
// d loss / d (visual LN parameters) given dlogits [n, C] of the n views whose activations vit_forward_saved holds.
// groups > 1: the n views belong to `groups` test samples (n / groups consecutive views each) and ln_grad is [groups, ln_count]
// wgrad_base (single sample only): also the gradient of every other visual parameter, into the flat e->vw_slots layout
static int vit_backward_ln(rlcf_engine* e, ClipModel& m, const float* feats, int n, const float* dlogits, float* ln_grad, hipStream_t st,
                           int groups = 1, float* wgrad_base = nullptr) {
    const rlcf_clip_cfg& c = m.cfg;
    const int Wv = c.vision_width, tok = m.tokens, T = n * tok, D = c.embed_dim, C = e->C, L = c.vision_layers;
    const int per = n / groups, gs = groups > 1 ? e->ln_count : 0;
    TRY(bwd_ensure(e, T, Wv));
    TRY(e->dfeat.ensure((size_t)e->max_views * D * sizeof(float))); TRY(e->dcls.ensure((size_t)e->max_views * Wv * sizeof(float)));
    RLCF_HIP_CHECK(hipMemsetAsync(ln_grad, 0, (size_t)groups * e->ln_count * sizeof(float), st));
    // d feat = scale * dlogits @ class_features  (logits = scale * feat @ class_features^T, custom_clip.py:429-430)
    TRY(launch_dimg(dlogits, e->txt0.as<float>(), n, C, D, m.logit_scale_exp, e->dfeat.as<float>(), st));
    TRY(launch_l2norm_bwd(feats, e->dfeat.as<float>(), e->vit_inv_norm.as<float>(), e->dfeat.as<float>(), n, D, st));
    if (wgrad_base)                        // visual.proj [Wv, D]: feat = ln_post(cls) @ proj  (model.py:237-238)
        TRY(wgrad(e, e->cls_ln.as<float>(), Wv, Wv, e->dfeat.as<float>(), D, D, n, wgrad_base + e->vw_slots[2].off, nullptr, st));
    TRY(gemm(e, e->dfeat.as<float>(), D, m.vproj, D, nullptr, nullptr, 0, nullptr, 0, e->dcls.as<float>(), Wv, n, Wv, D, 1.f, RLCF_EPI_NONE, st));
    float* gpost = ln_grad + (size_t)(2 + 4 * L) * Wv;
    const LnRef gpw = ln_ref(e, m.lnpost_w, 1);
    TRY(e->parts_ws.ensure(RLCF_PARTS_WS_FLOATS * sizeof(float)));
    TRY(launch_layernorm_bwd(e->cls_rows.as<float>(), gpw.p, e->dcls.as<float>(), nullptr, e->dcls.as<float>(), gpost, gpost + Wv, n, Wv, st,
                             groups > 1 ? per : 0, gs, groups > 1 ? gpw.group_stride : 0, PARTS_WS(e)));
    RLCF_HIP_CHECK(hipMemsetAsync(e->dX.p, 0, (size_t)T * Wv * sizeof(float), st));
    TRY(launch_scatter_rows(e->dcls.as<float>(), e->cls_row_idx.as<int32_t>(), e->dX.as<float>(), n, Wv, st));
    TRY(transformer_backward(e, m.vis, e->vt, e->vit_seqs.as<rlcf_seq>(), n, tok, (long)n * tok * tok, 0, T, st, ln_grad, tok,
                             groups > 1 ? per * tok : 0, gs, wgrad_base));
    if (!wgrad_base) {
        TRY(launch_vit_assemble_bwd(e->patch_out.as<float>(), m.cls, m.vpos, e->dX.as<float>(), ln_grad, ln_grad + Wv, n, tok, Wv, st,
                                    groups > 1 ? per : 0, gs, PARTS_WS(e)));
        return RLCF_OK;
    }
    // through ln_pre into the embedding (model.py:224-229): pre = [class_embedding | conv1(patches)] + positional_embedding
    float *pre = e->dH.as<float>(), *dpre = e->dA.as<float>(), *dpatch = e->dF.as<float>();      // backward scratch, free by now
    const int G2 = tok - 1, K = 3 * m.cfg.vision_patch_size * m.cfg.vision_patch_size;
    TRY(launch_vit_preln(e->patch_out.as<float>(), m.cls, m.vpos, pre, n, tok, Wv, st));
    TRY(launch_layernorm_bwd(pre, m.lnpre_w, e->dX.as<float>(), nullptr, dpre, ln_grad, ln_grad + Wv, T, Wv, st, 0, 0, 0, PARTS_WS(e)));
    float* gpos = wgrad_base + e->vw_slots[1].off;
    TRY(launch_colsum(dpre, tok * Wv, n, tok * Wv, gpos, st, PARTS_WS(e)));                                     // d positional_embedding = sum over views
    RLCF_HIP_CHECK(hipMemcpyAsync(wgrad_base + e->vw_slots[0].off, gpos, Wv * sizeof(float), hipMemcpyDeviceToDevice, st));   // d class_embedding = its row 0
    for (int v = 0; v < n; ++v)
        RLCF_HIP_CHECK(hipMemcpyAsync(dpatch + (size_t)v * G2 * Wv, dpre + ((size_t)v * tok + 1) * Wv, (size_t)G2 * Wv * sizeof(float),
                                      hipMemcpyDeviceToDevice, st));
    // conv1.weight [Wv, 3*ps*ps] (stride == kernel convolution = patches @ W^T, model.py:224): the patch matrix of vit_forward_saved is still there
    TRY(wgrad(e, dpatch, Wv, Wv, e->patches.as<float>(), m.Kp, K, n * G2, wgrad_base + e->vw_slots[3].off, nullptr, st));
    return RLCF_OK;
}

// LayerNorm tuning of B test images per tower pass (one AdamW step): every sample starts from the same reset state, so the
// selection forward, the reward pass and the saved forward of the selected views run on all B samples at once; the backward
// keeps the LayerNorm gradients per sample (grouped reductions), AdamW updates B parameter sets, and the clean-view inference
// runs once per sample with its own adapted LayerNorms (tune_cls_rl.py:206-227).
static int tta_batch_ln_fused(rlcf_engine* e, const float* views, int B, int N, const rlcf_tta_args* a, float* final_logits, int32_t* top5,
                              hipStream_t st) {
    ClipModel& s = e->model[RLCF_STUDENT];
    const int C = e->C, K = a->sample_k, D = s.cfg.embed_dim;
    const int n_sel = n_selected(a, N), BN = B * N, BS = B * n_sel;
    const size_t img_elems = (size_t)3 * s.cfg.image_resolution * s.cfg.image_resolution;
    const size_t nb = (size_t)e->ln_count * sizeof(float), np = (size_t)e->ln_count;
    TRY(e->ln_feat.ensure((size_t)e->max_views * D * sizeof(float)));
    TRY(e->b_ln.ensure(B * nb)); TRY(e->b_ln_m.ensure(B * nb)); TRY(e->b_ln_v.ensure(B * nb)); TRY(e->b_ln_grad.ensure(B * nb));
    TRY(e->b_logits.ensure((size_t)B * C * sizeof(float)));
    e->last_flops = 0.0;
    const float* cls_feat = e->txt0.as<float>();
    RLCF_HIP_CHECK(hipMemcpyAsync(e->ln_params.p, e->ln_init.p, nb, hipMemcpyDeviceToDevice, st));
    // 1. selection on all B*N views (pristine LayerNorms), 2. reward features of the selected views
    TRY(engine_encode_image(e, RLCF_STUDENT, views, BN, e->img_feat.as<float>(), st));
    TRY(engine_logits(e, e->img_feat.as<float>(), BN, cls_feat, C, e->logits.as<float>(), st));
    TRY(launch_entropy_select_batched(e->logits.as<float>(), B, N, C, n_sel, e->entropy.as<float>(), e->sel_idx.as<int32_t>(), st));
    TRY(launch_gather_rows(views, (int)img_elems, e->sel_idx.as<int32_t>(), e->views_sel.as<float>(), (int)img_elems, BS, (int)img_elems, st));
    TRY(reward_encode(e, BS, s.cfg.image_resolution, nullptr, st));
    // reset state of every sample (custom_clip.py:456-458 + optimizer.load_state_dict)
    TRY(launch_broadcast_rows(e->ln_init.as<float>(), e->b_ln.as<float>(), (int)np, B, st));
    RLCF_HIP_CHECK(hipMemsetAsync(e->b_ln_m.p, 0, B * nb, st));
    RLCF_HIP_CHECK(hipMemsetAsync(e->b_ln_v.p, 0, B * nb, st));
    for (int j = 0; j < a->tta_steps; ++j) {
        // 3. forward with saved activations on the selected views (each sample under its own LayerNorms), loss per sample,
        // 4. backward with per-sample LayerNorm gradients, 5. AdamW step j+1 of every sample
        e->lng_base = e->b_ln.as<float>(); e->lng_views = n_sel;
        int rc = vit_forward_saved(e, s, e->views_sel.as<float>(), BS, e->ln_feat.as<float>(), st);
        if (rc == RLCF_OK) rc = engine_logits(e, e->ln_feat.as<float>(), BS, cls_feat, C, e->sel_logits.as<float>(), st);
        if (rc == RLCF_OK) rc = launch_reward_loss_bank(e->sel_logits.as<float>(), C, nullptr, B, n_sel, C, K, reward_bank(e), a->clipscore_weight,
                                                        a->flags, a->min_entropy_w, e->topk_idx.as<int32_t>(), nullptr, nullptr, nullptr,
                                                        e->dlogits.as<float>(), e->rl_stats.as<float>(), st);
        if (rc == RLCF_OK) rc = vit_backward_ln(e, s, e->ln_feat.as<float>(), BS, e->dlogits.as<float>(), e->b_ln_grad.as<float>(), st, B);
        e->lng_base = nullptr; e->lng_views = 1;
        TRY(rc);
        TRY(launch_grad_nonfinite(e->b_ln_grad.as<float>(), (int64_t)np, B, e->step_skip.as<int32_t>(), st));
        TRY(launch_adamw(e->b_ln.as<float>(), e->b_ln_grad.as<float>(), e->b_ln_m.as<float>(), e->b_ln_v.as<float>(), (int64_t)B * np, j + 1,
                         a->lr, a->beta1, a->beta2, a->eps, a->weight_decay, st, e->step_skip.as<int32_t>(), (int64_t)np));
    }
    // 6. clean-view inference of the B samples in one pass: view b reads LayerNorm set b (tune_cls_rl.py:219-221)
    float* fl = final_logits ? final_logits : e->b_logits.as<float>();
    for (int b = 0; b < B; ++b)
        RLCF_HIP_CHECK(hipMemcpyAsync(e->views_sel.as<float>() + (size_t)b * img_elems, views + (size_t)b * N * img_elems,
                                      img_elems * sizeof(float), hipMemcpyDeviceToDevice, st));
    e->lng_base = e->b_ln.as<float>(); e->lng_views = 1;
    int rc = engine_encode_image(e, RLCF_STUDENT, e->views_sel.as<float>(), B, e->sel_feat.as<float>(), st);
    e->lng_base = nullptr;
    TRY(rc);
    TRY(engine_logits(e, e->sel_feat.as<float>(), B, cls_feat, C, fl, st));
    TRY(launch_top5_batched(fl, B, C, top5, st));
    return RLCF_OK;
}

int engine_tta_batch_ln(rlcf_engine* e, const float* views, int count, int N, const rlcf_tta_args* a, float* final_logits, int32_t* top5,
                        hipStream_t st) {
    ClipModel& s = e->model[RLCF_STUDENT];
    if (e->C <= 0 || e->image_bank || e->n_rewards <= 0) { rlcf_set_error("class bank / reward model not set"); return RLCF_ERR_STATE; }
    RLCF_ARG_CHECK(N > 0 && N <= e->max_views && a->sample_k > 0 && a->sample_k <= 32 && a->sample_k <= e->C);
    const bool rn = is_resnet(s.cfg);         // BatchNorm tuning: the batch statistics couple one sample's views, samples run one by one
    RLCF_ARG_CHECK(rn || s.tokens <= 320);
    const size_t per = (size_t)N * 3 * s.cfg.image_resolution * s.cfg.image_resolution;
    const int n_sel = n_selected(a, N), Bmax = e->max_views / N;
    const bool fused = !rn && Bmax >= 2 && a->tta_steps >= 1 && !a->skip_final && n_sel > 0;
    double flops = 0.0;
    int i = 0;
    while (i < count) {
        const int B = fused ? std::min(Bmax, count - i) : 1;
        if (fused && B >= 2) {
            TRY(tta_batch_ln_fused(e, views + (size_t)i * per, B, N, a, final_logits ? final_logits + (size_t)i * e->C : nullptr, top5 + (size_t)i * 5, st));
        } else {
            rlcf_tta_out o{};
            o.top5 = top5 + (size_t)i * 5;
            o.final_logits = final_logits ? final_logits + (size_t)i * e->C : nullptr;
            TRY(engine_tta_sample_ln(e, views + (size_t)i * per, N, a, &o, st));
        }
        flops += e->last_flops;
        i += B;
    }
    e->last_flops = flops / count;
    return RLCF_OK;
}

// ------------------------------------------------------------------ BatchNorm tuning of a ModifiedResNet student
// One iteration of tune_cls_rl.py:183-256 with CLIPCLS_TTA(arch=RN*, only_norm=True): the tuned tensors are the weight / bias of every
// BatchNorm2d whose name contains 'bn' (custom_clip.py:481-485: the downsample BatchNorms stay frozen).  The tuning passes run the
// BatchNorms on BATCH statistics (nn.BatchNorm2d in train mode, running statistics updated, or `_modified_bn_forward` under
// --prior_strength >= 0, tune_cls_rl.py:35-44), step 0 over all N views (the selection reads its logits; the statistics couple the
// views, so the backward covers all N with zero logit gradients outside the selection), later steps over the selected views.
// CLIPCLS_TTA.train() (custom_clip.py:487-497) puts the norm layers in train mode whatever `mode` is, so the final clean-view
// inference ALSO normalises with batch statistics (of that one image): reproduced.  The running statistics the sample leaves behind
// stay in e->bn_stats (rlcf_engine_get_bn_stats) until the next sample resets them.
static int rn_visual_reset(rlcf_engine* e, hipStream_t st) {   // visual.load_state_dict(initial_state_dict) for the flat buffer of a ResNet student
    if (!e->vw_count || !e->vw_dirty) return RLCF_OK;
    RLCF_HIP_CHECK(hipMemcpyAsync(e->vw.p, e->vw_init.p, e->vw_count * sizeof(float), hipMemcpyDeviceToDevice, st));
    e->vw_dirty = false;
    return rn_visual_refresh(e, st, e->vw_init_is_ckpt);
}
// full: every visual parameter is tuned (CLIPCLS_TTA(only_norm=False) on a ModifiedResNet — the parser defaults of tune_cls_rl.py,
// TPT/params.py:23,73): convolution / downsample.1 / attention-pool gradients into e->vw_grad, a second AdamW launch, the derived
// weight forms rebuilt after every step, and the final clean-view inference with the BatchNorms in EVAL form on the running statistics
// the tuning passes left behind (model.eval() is plain nn.Module.eval() when only_norm is off, custom_clip.py:487-497)
int engine_tta_sample_bn(rlcf_engine* e, const float* views, int N, const rlcf_tta_args* a, const rlcf_tta_out* out, hipStream_t st, bool full) {
    ClipModel& s = e->model[RLCF_STUDENT];
    TRY(engine_bn_enable(e, st));
    if (full) TRY(engine_rn_visual_enable(e, st));
    if (!full && s.rn.full_enabled && e->vw_dirty) TRY(rn_visual_reset(e, st));
    const size_t vb = full ? e->vw_count * sizeof(float) : 0;
    const int C = e->C, K = a->sample_k, D = s.cfg.embed_dim;
    const int n_sel = n_selected(a, N), n_e = n_sel * K;
    const size_t img_elems = (size_t)3 * s.cfg.image_resolution * s.cfg.image_resolution;
    const size_t nb = (size_t)e->ln_count * sizeof(float);
    const rlcf_tta_out none{};
    if (!out) out = &none;
    TRY(e->ln_feat.ensure((size_t)e->max_views * D * sizeof(float)));
    TRY(e->dfeat.ensure((size_t)e->max_views * D * sizeof(float)));
    TRY(e->bn_dlog.ensure((size_t)N * C * sizeof(float)));
    e->last_flops = 0.0;
    // model.reset(): visual.load_state_dict(initial_state_dict) restores parameters AND buffers (running statistics)
    RLCF_HIP_CHECK(hipMemcpyAsync(e->ln_params.p, e->ln_init.p, nb, hipMemcpyDeviceToDevice, st));
    RLCF_HIP_CHECK(hipMemcpyAsync(e->bn_stats.p, e->bn_stats_init.p, (size_t)s.rn.n_stats * sizeof(float), hipMemcpyDeviceToDevice, st));
    RLCF_HIP_CHECK(hipMemsetAsync(e->ln_m.p, 0, nb, st));
    RLCF_HIP_CHECK(hipMemsetAsync(e->ln_v.p, 0, nb, st));
    if (full) {
        TRY(rn_visual_reset(e, st));
        RLCF_HIP_CHECK(hipMemsetAsync(e->vw_m.p, 0, vb, st));
        RLCF_HIP_CHECK(hipMemsetAsync(e->vw_v.p, 0, vb, st));
    }
    const float* cls_feat = e->txt0.as<float>();
    for (int j = 0; j < a->tta_steps; ++j) {
        const int n = j == 0 ? N : n_sel;
        TRY(rn_forward_train(e, s, j == 0 ? views : e->views_sel.as<float>(), n, e->ln_feat.as<float>(), st));
        const float* dlog = e->dlogits.as<float>();
        if (j == 0) {
            TRY(engine_logits(e, e->ln_feat.as<float>(), N, cls_feat, C, e->logits.as<float>(), st));
            TRY(launch_entropy_select(e->logits.as<float>(), N, C, n_sel, e->entropy.as<float>(), e->sel_idx.as<int32_t>(), st));
            TRY(launch_gather_rows(views, (int)img_elems, e->sel_idx.as<int32_t>(), e->views_sel.as<float>(), (int)img_elems, n_sel, (int)img_elems, st));
            TRY(reward_encode(e, n_sel, s.cfg.image_resolution, out->reward_image_features, st));
            TRY(launch_gather_rows(e->logits.as<float>(), C, e->sel_idx.as<int32_t>(), e->sel_logits.as<float>(), C, n_sel, C, st));
            COPY_OUT(out->logits, e->logits.p, (size_t)N * C * sizeof(float));
            COPY_OUT(out->entropy, e->entropy.p, N * sizeof(float));
            COPY_OUT(out->selected_idx, e->sel_idx.p, n_sel * sizeof(int32_t));
        } else {
            TRY(engine_logits(e, e->ln_feat.as<float>(), n_sel, cls_feat, C, e->sel_logits.as<float>(), st));
        }
        TRY(launch_reward_loss_bank(e->sel_logits.as<float>(), C, nullptr, 1, n_sel, C, K, reward_bank(e),
                               a->clipscore_weight, a->flags, a->min_entropy_w, e->topk_idx.as<int32_t>(), e->clip_score.as<float>(),
                               e->rewards.as<float>(), e->loss.as<float>(), e->dlogits.as<float>(), e->rl_stats.as<float>(), st));
        if (j == 0) {
            RLCF_HIP_CHECK(hipMemsetAsync(e->bn_dlog.p, 0, (size_t)N * C * sizeof(float), st));
            TRY(launch_scatter_rows(e->dlogits.as<float>(), e->sel_idx.as<int32_t>(), e->bn_dlog.as<float>(), n_sel, C, st));
            dlog = e->bn_dlog.as<float>();
        }
        // d feat = scale * dlogits @ class_features (custom_clip.py:429-430), then the tower's backward down to the stem's first BatchNorm
        TRY(launch_dimg(dlog, cls_feat, n, C, D, s.logit_scale_exp, e->dfeat.as<float>(), st));
        if (full) RLCF_HIP_CHECK(hipMemsetAsync(e->vw_grad.p, 0, vb, st));
        TRY(rn_backward_bn(e, s, n, e->ln_feat.as<float>(), e->dfeat.as<float>(), e->ln_grad.as<float>(), st, full ? e->vw_grad.as<float>() : nullptr));
        if (j == 0) {
            if (full) COPY_OUT(out->vis_grad, e->vw_grad.p, vb);
            COPY_OUT(out->topk_idx, e->topk_idx.p, (size_t)n_e * sizeof(int32_t));
            COPY_OUT(out->clip_score, e->clip_score.p, (size_t)n_e * sizeof(float));
            COPY_OUT(out->rewards, e->rewards.p, (size_t)n_e * sizeof(float));
            COPY_OUT(out->loss, e->loss.p, sizeof(float));
            COPY_OUT(out->dlogits, e->dlogits.p, (size_t)n_sel * C * sizeof(float));
            COPY_OUT(out->ln_grad, e->ln_grad.p, nb);
        }
        TRY(launch_grad_nonfinite(e->ln_grad.as<float>(), e->ln_count, 1, e->step_skip.as<int32_t>(), st));
        if (full) TRY(launch_grad_nonfinite(e->vw_grad.as<float>(), (int64_t)e->vw_count, 1, e->step_skip.as<int32_t>(), st, true));
        if (out->step_skipped) COPY_OUT(out->step_skipped + j, e->step_skip.p, sizeof(int32_t));
        TRY(launch_adamw(e->ln_params.as<float>(), e->ln_grad.as<float>(), e->ln_m.as<float>(), e->ln_v.as<float>(), e->ln_count, j + 1,
                         a->lr, a->beta1, a->beta2, a->eps, a->weight_decay, st, e->step_skip.as<int32_t>(), e->ln_count));
        if (full) {
            TRY(launch_adamw(e->vw.as<float>(), e->vw_grad.as<float>(), e->vw_m.as<float>(), e->vw_v.as<float>(), (int64_t)e->vw_count, j + 1,
                             a->lr, a->beta1, a->beta2, a->eps, a->weight_decay, st, e->step_skip.as<int32_t>(), (int64_t)e->vw_count));
            e->vw_dirty = true;
            TRY(rn_visual_refresh(e, st));
        }
    }
    COPY_OUT(out->ln_after, e->ln_params.p, nb);
    if (full) COPY_OUT(out->vis_after, e->vw.p, vb);
    if (!a->skip_final) {
        // norm-layer tuning: the BatchNorms stay in train form (see the header comment); every-parameter tuning: eval form
        TRY(rn_forward_train(e, s, views, 1, e->img_feat.as<float>(), st, full ? 0 : -1));
        TRY(engine_logits(e, e->img_feat.as<float>(), 1, cls_feat, C, e->final_logits.as<float>(), st));
        TRY(launch_top5(e->final_logits.as<float>(), C, e->top5.as<int32_t>(), st));
        COPY_OUT(out->final_logits, e->final_logits.p, (size_t)C * sizeof(float));
        COPY_OUT(out->top5, e->top5.p, 5 * sizeof(int32_t));
    }
    RLCF_HIP_CHECK(hipMemcpyAsync(e->ln_params.p, e->ln_init.p, nb, hipMemcpyDeviceToDevice, st));
    if (full) TRY(rn_visual_reset(e, st));
    return RLCF_OK;
}

static int tta_sample_backbone(rlcf_engine* e, const float* views, int N, const rlcf_tta_args* a, const rlcf_tta_out* out, hipStream_t st, bool full);
int engine_tta_sample_ln(rlcf_engine* e, const float* views, int N, const rlcf_tta_args* a, const rlcf_tta_out* out, hipStream_t st) {
    return tta_sample_backbone(e, views, N, a, out, st, false);
}
// every visual parameter tuned (CLIPCLS_TTA only_norm=False, custom_clip.py:477-479)
int engine_tta_sample_visual(rlcf_engine* e, const float* views, int N, const rlcf_tta_args* a, const rlcf_tta_out* out, hipStream_t st) {
    TRY(engine_visual_enable(e, st));
    return tta_sample_backbone(e, views, N, a, out, st, true);
}
static int visual_reset(rlcf_engine* e, hipStream_t st) {      // visual.load_state_dict(initial_state_dict) for the flat buffer
    if (!e->vw_count || !e->vw_dirty) return RLCF_OK;
    RLCF_HIP_CHECK(hipMemcpyAsync(e->vw.p, e->vw_init.p, e->vw_count * sizeof(float), hipMemcpyDeviceToDevice, st));
    e->vw_dirty = false;
    return engine_visual_refresh(e, st, e->vw_init_is_ckpt);
}
static int tta_sample_backbone(rlcf_engine* e, const float* views, int N, const rlcf_tta_args* a, const rlcf_tta_out* out, hipStream_t st, bool full) {
    ClipModel& s = e->model[RLCF_STUDENT];
    if (e->C <= 0 || e->image_bank || e->n_rewards <= 0) { rlcf_set_error("class bank / reward model not set"); return RLCF_ERR_STATE; }
    RLCF_ARG_CHECK(N > 0 && N <= e->max_views && a && a->tta_steps >= 0 && a->sample_k > 0 && a->sample_k <= 32 && a->sample_k <= e->C);
    const int C = e->C, K = a->sample_k, D = s.cfg.embed_dim;
    const int n_sel = n_selected(a, N), n_e = n_sel * K;
    if (a->tta_steps > 0 && n_sel <= 0) { rlcf_set_error("int(N*selection_p) == 0 views selected (N=%d, p=%g)", N, a->selection_p); return RLCF_ERR_ARG; }
    if (is_resnet(s.cfg)) {
        return engine_tta_sample_bn(e, views, N, a, out, st, full);
    }
    RLCF_ARG_CHECK(s.tokens <= 320);
    const size_t img_elems = (size_t)3 * s.cfg.image_resolution * s.cfg.image_resolution;
    const size_t nb = (size_t)e->ln_count * sizeof(float);
    const rlcf_tta_out none{};
    if (!out) out = &none;
    TRY(e->ln_feat.ensure((size_t)e->max_views * D * sizeof(float)));
    e->last_flops = 0.0;
    // model.reset() (visual.load_state_dict(initial_state_dict), custom_clip.py:456-458) + optimizer state reset
    RLCF_HIP_CHECK(hipMemcpyAsync(e->ln_params.p, e->ln_init.p, nb, hipMemcpyDeviceToDevice, st));
    RLCF_HIP_CHECK(hipMemsetAsync(e->ln_m.p, 0, nb, st));
    RLCF_HIP_CHECK(hipMemsetAsync(e->ln_v.p, 0, nb, st));
    const size_t vb = full ? e->vw_count * sizeof(float) : 0;
    if (full) {
        TRY(visual_reset(e, st));
        RLCF_HIP_CHECK(hipMemsetAsync(e->vw_m.p, 0, vb, st));
        RLCF_HIP_CHECK(hipMemsetAsync(e->vw_v.p, 0, vb, st));
    }
    const float* cls_feat = e->txt0.as<float>();           // cached class text features (custom_clip.py:405-409)
    for (int j = 0; j < a->tta_steps; ++j) {
        if (j == 0) {
            // all N views decide the selection; only the selected ones carry gradient (rows outside idx get zero grad)
            TRY(engine_encode_image(e, RLCF_STUDENT, views, N, e->img_feat.as<float>(), st));
            TRY(engine_logits(e, e->img_feat.as<float>(), N, cls_feat, C, e->logits.as<float>(), st));
            TRY(launch_entropy_select(e->logits.as<float>(), N, C, n_sel, e->entropy.as<float>(), e->sel_idx.as<int32_t>(), st));
            if (a->flags & RLCF_F_NO_SELECTION) TRY(launch_iota(e->sel_idx.as<int32_t>(), n_sel, st));     // retrieval: every row, in order
            TRY(launch_gather_rows(views, (int)img_elems, e->sel_idx.as<int32_t>(), e->views_sel.as<float>(), (int)img_elems, n_sel, (int)img_elems, st));
            TRY(reward_encode(e, n_sel, s.cfg.image_resolution, out->reward_image_features, st));
            COPY_OUT(out->logits, e->logits.p, (size_t)N * C * sizeof(float));
            COPY_OUT(out->entropy, e->entropy.p, N * sizeof(float));
            COPY_OUT(out->selected_idx, e->sel_idx.p, n_sel * sizeof(int32_t));
        }
        TRY(vit_forward_saved(e, s, e->views_sel.as<float>(), n_sel, e->ln_feat.as<float>(), st));
        TRY(engine_logits(e, e->ln_feat.as<float>(), n_sel, cls_feat, C, e->sel_logits.as<float>(), st));
        TRY(launch_reward_loss_bank(e->sel_logits.as<float>(), C, nullptr, 1, n_sel, C, K, reward_bank(e),
                               a->clipscore_weight, a->flags, a->min_entropy_w, e->topk_idx.as<int32_t>(), e->clip_score.as<float>(),
                               e->rewards.as<float>(), e->loss.as<float>(), e->dlogits.as<float>(), e->rl_stats.as<float>(), st));
        if (full) RLCF_HIP_CHECK(hipMemsetAsync(e->vw_grad.p, 0, vb, st));
        TRY(vit_backward_ln(e, s, e->ln_feat.as<float>(), n_sel, e->dlogits.as<float>(), e->ln_grad.as<float>(), st, 1,
                            full ? e->vw_grad.as<float>() : nullptr));
        if (j == 0) {
            if (full) COPY_OUT(out->vis_grad, e->vw_grad.p, vb);
            COPY_OUT(out->topk_idx, e->topk_idx.p, (size_t)n_e * sizeof(int32_t));
            COPY_OUT(out->clip_score, e->clip_score.p, (size_t)n_e * sizeof(float));
            COPY_OUT(out->rewards, e->rewards.p, (size_t)n_e * sizeof(float));
            COPY_OUT(out->loss, e->loss.p, sizeof(float));
            COPY_OUT(out->dlogits, e->dlogits.p, (size_t)n_sel * C * sizeof(float));
            COPY_OUT(out->ln_grad, e->ln_grad.p, nb);
        }
        // one optimizer over both buffers: an inf / NaN anywhere skips the whole step (GradScaler.step, tpt_cls_rl.py:78)
        TRY(launch_grad_nonfinite(e->ln_grad.as<float>(), e->ln_count, 1, e->step_skip.as<int32_t>(), st));
        if (full) TRY(launch_grad_nonfinite(e->vw_grad.as<float>(), (int64_t)e->vw_count, 1, e->step_skip.as<int32_t>(), st, true));
        if (out->step_skipped) COPY_OUT(out->step_skipped + j, e->step_skip.p, sizeof(int32_t));
        TRY(launch_adamw(e->ln_params.as<float>(), e->ln_grad.as<float>(), e->ln_m.as<float>(), e->ln_v.as<float>(), e->ln_count, j + 1,
                         a->lr, a->beta1, a->beta2, a->eps, a->weight_decay, st, e->step_skip.as<int32_t>(), e->ln_count));
        if (full) {
            TRY(launch_adamw(e->vw.as<float>(), e->vw_grad.as<float>(), e->vw_m.as<float>(), e->vw_v.as<float>(), (int64_t)e->vw_count, j + 1,
                             a->lr, a->beta1, a->beta2, a->eps, a->weight_decay, st, e->step_skip.as<int32_t>(), (int64_t)e->vw_count));
            e->vw_dirty = true;
            TRY(engine_visual_refresh(e, st));
        }
    }
    COPY_OUT(out->ln_after, e->ln_params.p, nb);
    if (full) COPY_OUT(out->vis_after, e->vw.p, vb);
    if (!a->skip_final) {
        // final clean-view inference with the adapted LayerNorms (tune_cls_rl.py:219-221)
        TRY(engine_encode_image(e, RLCF_STUDENT, views, 1, e->img_feat.as<float>(), st));
        TRY(engine_logits(e, e->img_feat.as<float>(), 1, cls_feat, C, e->final_logits.as<float>(), st));
        TRY(launch_top5(e->final_logits.as<float>(), C, e->top5.as<int32_t>(), st));
        COPY_OUT(out->final_logits, e->final_logits.p, (size_t)C * sizeof(float));
        COPY_OUT(out->top5, e->top5.p, 5 * sizeof(int32_t));
    }
    // leave the engine in its pristine state for the prompt path (which assumes frozen, pristine weights)
    RLCF_HIP_CHECK(hipMemcpyAsync(e->ln_params.p, e->ln_init.p, nb, hipMemcpyDeviceToDevice, st));
    if (full) TRY(visual_reset(e, st));
    return RLCF_OK;
}
