"""Post-adaptation evaluation driver (reference src/evaluation/utils.py:9-66): MCQ + retrieval metrics, one log
record, one line appended to ``<checkpoint_path>/results.jsonl``.  wandb / tensorboard are optional hooks here (the
reference imports wandb unconditionally); nothing in this module touches the GPU itself."""
import json
import logging
import os

from .mcq import mcq_eval
from .retrieval import retrieval_eval


def evaluate(model, data, epoch, args, score_matrix_t2i=None, tb_writer=None, tokenizer=None):
    metrics = {}
    model.eval()
    metrics.update(mcq_eval(model, data, epoch, args, tokenizer=tokenizer))
    metrics.update(retrieval_eval(model, data, args, score_matrix_t2i, tokenizer=tokenizer))
    if not metrics:
        return metrics
    logging.info(f"Eval Epoch: {epoch}\n " + "\n".join(
        f"{k}: {round(v, 4):.4f}" if isinstance(v, (int, float)) else f"{k}: {v}" for k, v in metrics.items()))
    if getattr(args, "save_logs", False):
        if tb_writer is not None:
            for name, val in metrics.items():
                if isinstance(val, (int, float)):
                    tb_writer.add_scalar("val/" + name, val, epoch)
        with open(os.path.join(args.checkpoint_path, "results.jsonl"), "a+") as f:
            f.write(json.dumps(metrics))
            f.write("\n")
    if getattr(args, "wandb", False):
        import wandb                                              # optional, only when asked for
        log_data = {"val/" + name: val for name, val in metrics.items()}
        log_data["epoch"] = epoch
        wandb.log(log_data)
    return metrics


evaluate_video = evaluate    # reference evaluate_video (:69-125) differs only in the dataset names, chosen by args.video
