"""Test-set metrics the reference drivers print after training (`score()`, reference mfm_mosi.py:483-499 for the
regression datasets, mfm_you.py:556-564 for the classification ones) -- restated on numpy so that the package has
no scikit-learn dependency.  Definitions follow sklearn.metrics (what the reference calls): weighted F1 =
support-weighted mean of per-class F1 over the labels present in y_true or y_pred; confusion matrix rows = true
labels, columns = predicted, labels sorted; the classification report is sklearn's text layout (digits=5).

NOTE the reference passes `f1_score(np.round(predictions), np.round(y_test))`, i.e. the PREDICTIONS as y_true:
`score()` below keeps that argument order so its numbers equal the reference's."""
import sys

import numpy as np


def _labels(a, b):
    return np.unique(np.concatenate([np.asarray(a).ravel(), np.asarray(b).ravel()]))


def confusion_matrix(y_true, y_pred, labels=None):
    y_true, y_pred = np.asarray(y_true).ravel(), np.asarray(y_pred).ravel()
    labels = _labels(y_true, y_pred) if labels is None else np.asarray(labels)
    idx = {l: i for i, l in enumerate(labels.tolist())}
    cm = np.zeros((len(labels), len(labels)), dtype=np.int64)
    for t, p in zip(y_true.tolist(), y_pred.tolist()):
        if t in idx and p in idx:
            cm[idx[t], idx[p]] += 1
    return cm


def precision_recall_f1_support(y_true, y_pred, labels=None):
    labels = _labels(y_true, y_pred) if labels is None else np.asarray(labels)
    cm = confusion_matrix(y_true, y_pred, labels)
    tp = np.diag(cm).astype(np.float64)
    pred_n, true_n = cm.sum(0).astype(np.float64), cm.sum(1).astype(np.float64)
    with np.errstate(divide="ignore", invalid="ignore"):
        prec = np.where(pred_n > 0, tp / pred_n, 0.0)
        rec = np.where(true_n > 0, tp / true_n, 0.0)
        f1 = np.where(prec + rec > 0, 2 * prec * rec / (prec + rec), 0.0)
    return prec, rec, f1, true_n.astype(np.int64), labels


def f1_score(y_true, y_pred, average="weighted"):
    prec, rec, f1, sup, _ = precision_recall_f1_support(y_true, y_pred)
    if average == "weighted":
        return float((f1 * sup).sum() / max(sup.sum(), 1))
    if average == "macro":
        return float(f1.mean())
    raise ValueError("average must be 'weighted' or 'macro'")


def accuracy_score(y_true, y_pred):
    return float(np.mean(np.asarray(y_true).ravel() == np.asarray(y_pred).ravel()))


def classification_report(y_true, y_pred, digits=5):
    prec, rec, f1, sup, labels = precision_recall_f1_support(y_true, y_pred)
    names = [str(l) for l in labels.tolist()]
    width = max(max(len(n) for n in names), len("weighted avg"), digits)
    head = "{:>{w}s} ".format("", w=width) + " ".join("{:>9}".format(h) for h in ("precision", "recall", "f1-score", "support"))
    lines = [head, ""]
    row = "{:>{w}s} " + " {:>9.{d}f}" * 3 + " {:>9}"
    for n, p, r, f, s in zip(names, prec, rec, f1, sup):
        lines.append(row.format(n, p, r, f, int(s), w=width, d=digits))
    lines.append("")
    tot = int(sup.sum())
    acc = accuracy_score(y_true, y_pred)
    lines.append("{:>{w}s} ".format("accuracy", w=width) + " {:>9} {:>9}".format("", "") + " {:>9.{d}f} {:>9}".format(acc, tot, d=digits))
    lines.append(row.format("macro avg", prec.mean(), rec.mean(), f1.mean(), tot, w=width, d=digits))
    wsum = max(sup.sum(), 1)
    lines.append(row.format("weighted avg", (prec * sup).sum() / wsum, (rec * sup).sum() / wsum, (f1 * sup).sum() / wsum,
                            tot, w=width, d=digits))
    return "\n".join(lines) + "\n"


def score(predictions, y_test, out=sys.stdout):
    """mfm_mosi.py:483-499, line for line (regression datasets: MOSI, MMMO, MOUD ...)."""
    predictions, y_test = np.asarray(predictions, dtype=np.float64), np.asarray(y_test, dtype=np.float64)
    res = {}
    res["mae"] = float(np.mean(np.absolute(predictions - y_test)))
    print("mae: ", res["mae"], file=out)
    res["corr"] = float(np.corrcoef(predictions, y_test)[0][1])
    print("corr: ", res["corr"], file=out)
    res["mult_acc"] = round(float(np.sum(np.round(predictions) == np.round(y_test))) / float(len(y_test)), 5)
    print("mult_acc: ", res["mult_acc"], file=out)
    res["mult_f_score"] = round(f1_score(np.round(predictions), np.round(y_test), average="weighted"), 5)
    print("mult f_score: ", res["mult_f_score"], file=out)
    true_label = (y_test >= 0)
    predicted_label = (predictions >= 0)
    print("Confusion Matrix :", file=out)
    res["confusion"] = confusion_matrix(true_label, predicted_label)
    print(res["confusion"], file=out)
    print("Classification Report :", file=out)
    print(classification_report(true_label, predicted_label, digits=5), file=out)
    res["accuracy"] = accuracy_score(true_label, predicted_label)
    print("Accuracy ", res["accuracy"], file=out)
    out.flush()
    return res


def score_classes(logits, y_test, out=sys.stdout):
    """mfm_you.py:556-564, line for line (classification datasets: YouTube, POM ...): confusion matrix, report and
    accuracy over the arg-max class."""
    pred = np.argmax(np.asarray(logits), axis=1)
    y_test = np.asarray(y_test).astype(np.int64)
    res = {"accuracy": accuracy_score(y_test, pred), "confusion": confusion_matrix(y_test, pred)}
    print("Confusion Matrix :", file=out)
    print(res["confusion"], file=out)
    print("Classification Report :", file=out)
    print(classification_report(y_test, pred, digits=5), file=out)
    print("Accuracy ", res["accuracy"], file=out)
    out.flush()
    return res
