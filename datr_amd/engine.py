"""One burn-in training epoch: the counterpart of the reference's `engine.train_one_epoch`
(/root/reference/engine.py:29-142) with the same signature and the same per-step sequence:
H2D move, model(samples, targets), criterion, sum(loss * weight) over weight_dict keys,
reduce_dict for logging, non-finite guard, zero_grad / backward / clip_grad_norm_ / step.
The reference's MetricLogger pretty-printer is bookkeeping (out of scope); the
returned dict carries the same averaged stats keys (`loss`, every `<key>` scaled and
`<key>_unscaled`, `class_error`, `lr`)."""
from __future__ import annotations

import math
import sys
import time
from collections import defaultdict
from typing import Iterable

import torch

from .criterion import weighted_total
from .dist import reducer_for
from .nested import reduce_dict


class _LossFetch:
    """The reduced loss dict on its way to the host: ONE stacked device tensor copied into pinned
    memory without blocking (the reference calls .item() / float() per entry: ~170 synchronising
    copies per step).  `result()` waits for that copy only -- it is called after backward, clip
    and optimizer step have been enqueued, so the host never drains the GPU's queue mid-step.
    The non-finite guard it feeds ends the process (engine.py:81-84), so whether the dying
    process had already enqueued its update is unobservable; the values are the reference's."""

    def __init__(self, loss_dict_reduced, weight_dict):
        self.keys = list(loss_dict_reduced)
        self.weight_dict = weight_dict
        self.event = None
        vals = [loss_dict_reduced[k] for k in self.keys]
        if self.keys and all(torch.is_tensor(v) for v in vals):
            dev = torch.stack([v.detach().float().reshape(()) for v in vals])
            if dev.is_cuda:
                self.host = torch.empty(dev.shape, dtype=dev.dtype, pin_memory=True)
                self.host.copy_(dev, non_blocking=True)
                self.event = torch.cuda.Event()
                self.event.record()
            else:
                self.host = dev
        else:
            self.host = [float(v) for v in vals]

    def result(self):
        """({key: loss * weight} for weighted keys, {key: loss}) as python floats."""
        if self.event is not None:
            self.event.synchronize()
        host = self.host.tolist() if torch.is_tensor(self.host) else self.host
        unscaled = dict(zip(self.keys, host))
        scaled = {k: v * float(self.weight_dict[k]) for k, v in unscaled.items()
                  if k in self.weight_dict}
        return scaled, unscaled


def _losses_to_host(loss_dict_reduced, weight_dict):
    return _LossFetch(loss_dict_reduced, weight_dict).result()


def _check_finite(loss_value, loss_dict_reduced):
    if not math.isfinite(loss_value):
        print(f"Loss is {loss_value}, stopping training")
        print(loss_dict_reduced)
        sys.exit(1)


_GC_FROZEN = [False, 0]         # done?, training steps seen by this process

# Diagnostics (bench.py --gpus N): a list here receives one dict per training step with the host's side of
# it -- `host_enqueue_ms`: wall time until batch move + forward + criterion + backward + reduce + clip + step are
# ENQUEUED (NB it includes back-pressure: a host that runs ahead blocks inside launch calls once the runtime's
# queue is full, so in steady state this approaches the GPU's step time whoever is the bound), `host_cpu_ms`: CPU
# time of the PROCESS over the whole step (all threads: the forward's launches come from the calling thread, the
# backward's from the autograd engine's, and the HIP runtime's own threads poll -- measured ~1.9 cores' worth per
# rank: the number of cores a rank needs), `host_cpu_main_thread_ms`: the calling thread's share (forward,
# criterion, optimizer enqueue: the loop is host-bound when the busiest thread approaches the step time), `loss_wait_ms`: what the host then waits for the step's loss values (the one
# blocking point of the loop).  None = off: the clock reads are not taken.
STEP_DIAG = None


def _settle_garbage_collector(steps_done: int = 0) -> None:
    """Once the first steps have built the long-lived objects (modules, cached plans, index tensors, the
    optimizer's state), collect and FREEZE them (`gc.freeze`): a full collection then walks only what was
    created since.  Without it the cyclic collector's oldest generation comes round every ~50 steps and
    walks everything -- measured 80-90 ms of host time in one step, during which the device runs dry
    (tools/probes/monitor_hiccup2.py: steps of 162 ms among 76 ms ones; none with the collector off).
    This is a PROCESS-GLOBAL side effect of calling the epoch functions (objects alive at the third step are never
    examined by the cyclic collector again; reference counting still frees them): DATR_FREEZE_GC=0 switches it
    off."""
    import os
    _GC_FROZEN[1] += 1
    if _GC_FROZEN[1] == 3 and not _GC_FROZEN[0] and os.environ.get("DATR_FREEZE_GC", "1") != "0":
        import gc
        gc.collect()
        gc.freeze()
        _GC_FROZEN[0] = True


def _backward_and_step(model, optimizer, losses, max_norm, scaler, amp, reducer=None):
    """zero_grad / backward / gradient all-reduce / clip / step (engine.py:86-104).  With a
    reducer (datr_amd.dist.GradAllReducer: the counterpart of the reference's DDP wrapper,
    main.py:156) the gradients live in its flat buckets and are averaged over the ranks while
    backward is still running."""
    if reducer is not None:
        reducer.zero_grad()
    else:
        optimizer.zero_grad()
    own_step = hasattr(optimizer, "clip_and_step") and not amp
    if amp:
        scaler.scale(losses).backward()
        if reducer is not None:
            reducer.finish()
            reducer.drop_unused_grads()
        if max_norm > 0:
            scaler.unscale_(optimizer)
            torch.nn.utils.clip_grad_norm_(model.parameters(), max_norm)
        scaler.step(optimizer)
        scaler.update()
    else:
        losses.backward()
        if reducer is not None:
            reducer.finish()
        if own_step:
            # clip + AdamW as multi-tensor launches of the own kernels (datr_amd.optim): the clip
            # coefficient and the reducer's "some rank used this parameter" flags stay on the device
            used = None
            if reducer is not None:
                if getattr(optimizer, "_index", None) is None:
                    optimizer.set_used_order(reducer.params)
                used = reducer.used_flags()
            optimizer.clip_and_step(max_norm, used)
            return
        if reducer is not None:
            reducer.drop_unused_grads()
        if max_norm > 0:
            torch.nn.utils.clip_grad_norm_(model.parameters(), max_norm)
        optimizer.step()


def train_one_epoch(model: torch.nn.Module, criterion: torch.nn.Module, data_loader: Iterable,
                    optimizer: torch.optim.Optimizer, device: torch.device, epoch: int,
                    max_norm: float = 0, wo_class_error=False, lr_scheduler=None, args=None,
                    logger=None, ema_m=None):
    amp = bool(getattr(args, "amp", False))
    scaler = torch.amp.GradScaler("cuda", enabled=amp)
    need_tgt_for_training = bool(getattr(args, "use_dn", False))
    reducer = reducer_for(model, args)
    model.train()
    criterion.train()
    sums, counts = defaultdict(float), defaultdict(int)
    last = {}
    steps = 0
    for samples, targets, _, _ in data_loader:
        diag = STEP_DIAG
        t_begin = time.perf_counter() if diag is not None else 0.0
        c_begin = (time.process_time(), time.thread_time()) if diag is not None else (0.0, 0.0)
        samples = samples.to(device)
        targets = [{k: v.to(device) for k, v in t.items()} for t in targets]
        if hasattr(criterion, "prefetch_num_boxes"):
            criterion.prefetch_num_boxes(targets, device)
        with torch.autocast(device_type=device.type, enabled=amp):
            outputs = model(samples, targets) if need_tgt_for_training else model(samples)
            loss_dict = criterion(outputs, targets)
            weight_dict = criterion.weight_dict
            losses = weighted_total(loss_dict, weight_dict)

        loss_dict_reduced = {k: v.detach() for k, v in reduce_dict(loss_dict).items()}
        fetch = _LossFetch(loss_dict_reduced, weight_dict)
        _backward_and_step(model, optimizer, losses, max_norm, scaler, amp, reducer)
        t_enqueued = time.perf_counter() if diag is not None else 0.0
        scaled, unscaled = fetch.result()
        if diag is not None:
            diag.append({"host_enqueue_ms": (t_enqueued - t_begin) * 1e3,
                         "loss_wait_ms": (time.perf_counter() - t_enqueued) * 1e3,
                         "host_cpu_ms": (time.process_time() - c_begin[0]) * 1e3,
                         "host_cpu_main_thread_ms": (time.thread_time() - c_begin[1]) * 1e3})
        loss_value = sum(scaled.values())
        _check_finite(loss_value, loss_dict_reduced)
        if getattr(args, "onecyclelr", False):
            lr_scheduler.step()
        if getattr(args, "use_ema", False) and epoch >= getattr(args, "ema_epoch", 0):
            ema_m.update(model)

        stats = {"loss": loss_value, "lr": optimizer.param_groups[0]["lr"]}
        stats.update(scaled)
        stats.update({f"{k}_unscaled": v for k, v in unscaled.items()})
        if "class_error" in unscaled:
            stats["class_error"] = unscaled["class_error"]
        for k, v in stats.items():
            sums[k] += v
            counts[k] += 1
        last = stats
        steps += 1
        _settle_garbage_collector(steps)
        if getattr(args, "debug", False) and steps % 15 == 0:
            print("BREAK!" * 5)
            break
    resstat = {k: sums[k] / counts[k] for k in sums if counts[k] > 0}
    resstat["_last"] = last
    return resstat


def train_one_epoch_with_self_training(model, teacher_model, criterion, data_loader,
                                       data_loader_strong_aug, optimizer, device, epoch,
                                       max_norm: float = 0, wo_class_error=False, lr_scheduler=None,
                                       args=None, logger=None, ema_m=None):
    """Teacher-student epoch: counterpart of the reference's
    `engine.train_one_epoch_with_self_training` (/root/reference/engine.py:146-342).  Per step:
    the EMA teacher (eval mode) predicts the weakly-augmented target images -> PostProcess with
    num_select=100 on unit sizes -> per-class threshold -> class-aware NMS(0.7)[:100] ->
    pseudo targets; the student sees source + strongly-augmented target images with
    self_training_flag=True; loss = source losses + loss_self_training * target losses."""
    import numpy as np

    from .detector import PostProcess
    from .self_training import (deal_pesudo_label, get_pseudo_label_via_threshold,
                                get_unlabel_img, get_valid_output, rescale_pseudo_targets,
                                spilt_output)
    amp = bool(getattr(args, "amp", False))
    scaler = torch.amp.GradScaler("cuda", enabled=amp)
    need_tgt_for_training = bool(getattr(args, "use_dn", False))
    model.train()
    criterion.train()
    reducer = reducer_for(model, args)
    post = PostProcess()                      # default-constructed: num_select = 100 (engine.py:159)
    loader = data_loader_strong_aug if data_loader_strong_aug is not None else data_loader
    sums, counts = defaultdict(float), defaultdict(int)
    last, steps = {}, 0
    for samples, source_labels, target_labels, samples_strong_aug in loader:
        samples = samples.to(device)
        source_labels = [{k: v.to(device) for k, v in t.items()} for t in source_labels]
        if samples_strong_aug is not None:
            samples_strong_aug = samples_strong_aug.to(device)
        unlabel_img = get_unlabel_img(samples)
        with torch.no_grad():
            teacher_out = teacher_model.ema(unlabel_img)
        target_labels = [{k: v.to(device) for k, v in t.items()} for t in target_labels]
        unit = torch.ones(len(target_labels), 2, dtype=torch.long, device=device)
        results = post(teacher_out, unit, not_to_xyxy=True)

        with torch.autocast(device_type=device.type, enabled=amp):
            # The student forward and the source-domain criterion do not depend on the pseudo
            # labels: they are ENQUEUED before the pseudo-label selection below, whose
            # data-dependent sizes force host synchronisations (threshold, NMS) -- the GPU then
            # works through the student forward while the host waits, instead of idling after
            # the teacher forward (the reference selects first, engine.py:213-226; same values).
            if need_tgt_for_training:
                outputs = model(samples_strong_aug, source_labels, self_training_flag=True)
            else:
                outputs = model(samples_strong_aug, self_training_flag=True)
            source_outputs, target_outputs = spilt_output(outputs)
            weight_dict = criterion.weight_dict
            loss_dict_source = criterion(source_outputs, source_labels, target_domain_flag=False)

        threshold = np.asarray([args.pseudo_label_threshold] * args.num_classes)
        idx_list, labels_d, boxes_d, scores_d = get_pseudo_label_via_threshold(results, threshold=threshold)
        pseudo = deal_pesudo_label(target_labels, idx_list, labels_d, boxes_d, scores_d)
        pseudo = rescale_pseudo_targets(unlabel_img, pseudo)

        with torch.autocast(device_type=device.type, enabled=amp):
            valid_target_outputs, pseudo_list = get_valid_output(target_outputs, pseudo, idx_list)
            loss_dict_target = criterion(valid_target_outputs, pseudo_list, target_domain_flag=True)
            losses_source = weighted_total(loss_dict_source, weight_dict)
            losses_target = weighted_total(loss_dict_target, weight_dict)
            if isinstance(losses_target, int) and losses_target == 0:
                losses_target = torch.tensor(0)
            losses = losses_source + losses_target * weight_dict["loss_self_training"]

        loss_dict_reduced = reduce_dict(loss_dict_source)
        fetch = _LossFetch(loss_dict_reduced, weight_dict)
        _backward_and_step(model, optimizer, losses, max_norm, scaler, amp, reducer)
        scaled, unscaled = fetch.result()
        loss_value = sum(scaled.values())
        _check_finite(loss_value, loss_dict_reduced)
        if getattr(args, "onecyclelr", False):
            lr_scheduler.step()
        if getattr(args, "use_ema", False) and epoch >= getattr(args, "ema_epoch", 0):
            ema_m.update(model)
        stats = {"loss": loss_value, "lr": optimizer.param_groups[0]["lr"],
                 "loss_self_training_sum": float(losses_target.detach()),
                 "num_pseudo_images": float(len(idx_list))}
        stats.update(scaled)
        stats.update({f"{k}_unscaled": v for k, v in unscaled.items()})
        if "class_error" in unscaled:
            stats["class_error"] = unscaled["class_error"]
        for k, v in stats.items():
            sums[k] += v
            counts[k] += 1
        last = dict(stats, total_loss=float(losses.detach()),
                    target_loss_dict=_losses_to_host(loss_dict_target, {})[1],
                    pseudo_targets=pseudo_list)
        steps += 1
        _settle_garbage_collector(steps)
        if getattr(args, "debug", False) and steps % 15 == 0:
            print("BREAK!" * 5)
            break
    resstat = {k: sums[k] / counts[k] for k in sums if counts[k] > 0}
    resstat["_last"] = last
    return resstat


@torch.no_grad()
def evaluate(model, criterion, postprocessors, data_loader, base_ds, device, output_dir=None,
             wo_class_error=False, args=None, logger=None):
    """Evaluation loop: counterpart of the reference's `engine.evaluate`
    (/root/reference/engine.py:349-523) for the 'bbox' post-processor.  Per batch
    `(samples, _, targets)`: eval-mode forward, criterion (for the logged losses), PostProcess on
    `orig_size`, results keyed by `image_id` into the evaluator; then cross-rank gather,
    accumulate, summarize.  `base_ds`: a COCO-format ground-truth dict, or None to take the
    ground truth from the loop's targets.  Returns (stats, evaluator) with
    stats['coco_eval_bbox'] = the 12 COCO numbers ([1] = mAP50), as the reference does."""
    from .evaluation import BoxEvaluator
    need_tgt_for_training = bool(getattr(args, "use_dn", False))
    amp = bool(getattr(args, "amp", False))
    model.eval()
    criterion.eval()
    evaluator = BoxEvaluator(base_ds, use_cats=bool(getattr(args, "useCats", True)))
    sums, counts = defaultdict(float), defaultdict(int)
    steps = 0
    for samples, _, targets in data_loader:
        samples = samples.to(device)
        targets = [{k: (v.to(device) if torch.is_tensor(v) else v) for k, v in t.items()} for t in targets]
        with torch.autocast(device_type=torch.device(device).type, enabled=amp):
            outputs = model(samples, targets) if need_tgt_for_training else model(samples)
            loss_dict = criterion(outputs, targets)
        weight_dict = criterion.weight_dict
        reduced = reduce_dict(loss_dict)
        scaled = {k: v * weight_dict[k] for k, v in reduced.items() if k in weight_dict}
        stats = {"loss": float(sum(scaled.values()))}
        stats.update({k: float(v) for k, v in scaled.items()})
        stats.update({f"{k}_unscaled": float(v) for k, v in reduced.items()})
        if "class_error" in reduced and not wo_class_error:
            stats["class_error"] = float(reduced["class_error"])
        for k, v in stats.items():
            sums[k] += v
            counts[k] += 1
        orig_target_sizes = torch.stack([t["orig_size"] for t in targets], dim=0)
        results = postprocessors["bbox"](outputs, orig_target_sizes)
        res = {int(t["image_id"].reshape(-1)[0]): r for t, r in zip(targets, results)}
        if base_ds is None:
            evaluator.add_ground_truth(targets)
        evaluator.update(res)
        steps += 1
        if getattr(args, "debug", False) and steps % 15 == 0:
            print("BREAK!" * 5)
            break
    evaluator.synchronize_between_processes()
    evaluator.accumulate()
    evaluator.summarize(verbose=logger is None)
    out = {k: sums[k] / counts[k] for k in sums if counts[k] > 0}
    out["coco_eval_bbox"] = list(evaluator.stats)
    return out, evaluator
