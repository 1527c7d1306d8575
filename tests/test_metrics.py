"""score() metrics (reference mfm_mosi.py:483-499, mfm_you.py:556-564) restated on numpy: against scikit-learn
(what the reference calls) when it is importable, and against hand-computed values always.  CPU only."""
import io

import numpy as np
import pytest

from factorized_amd import metrics as M


def test_hand_computed_values():
    yt = np.array([0, 0, 1, 1, 1, 2])
    yp = np.array([0, 1, 1, 1, 0, 2])
    cm = M.confusion_matrix(yt, yp)
    assert cm.tolist() == [[1, 1, 0], [1, 2, 0], [0, 0, 1]]
    prec, rec, f1, sup, labels = M.precision_recall_f1_support(yt, yp)
    assert np.allclose(prec, [0.5, 2 / 3, 1.0]) and np.allclose(rec, [0.5, 2 / 3, 1.0]) and sup.tolist() == [2, 3, 1]
    assert abs(M.f1_score(yt, yp) - (0.5 * 2 + (2 / 3) * 3 + 1.0) / 6) < 1e-12
    assert abs(M.accuracy_score(yt, yp) - 4 / 6) < 1e-12
    # a label that is only predicted gets support 0 and drops out of the weighted mean, but not of the macro mean
    assert abs(M.f1_score([0, 0], [0, 1]) - (2 / 3)) < 1e-12
    assert abs(M.f1_score([0, 0], [0, 1], average="macro") - (1 / 3)) < 1e-12


def test_against_sklearn_when_available():
    sk = pytest.importorskip("sklearn.metrics")
    rs = np.random.RandomState(0)
    for _ in range(5):
        y = rs.uniform(-3, 3, size=200)
        p = y + rs.normal(0, 1.0, size=200)
        a, b = np.round(p), np.round(y)
        assert abs(M.f1_score(a, b) - sk.f1_score(a, b, average="weighted")) < 1e-12
        tl, pl = y >= 0, p >= 0
        assert (M.confusion_matrix(tl, pl) == sk.confusion_matrix(tl, pl)).all()
        assert abs(M.accuracy_score(tl, pl) - sk.accuracy_score(tl, pl)) < 1e-12
        assert M.classification_report(tl, pl, digits=5).split() == sk.classification_report(tl, pl, digits=5).split()
        yc = rs.randint(0, 3, size=100)
        pc = rs.randint(0, 3, size=100)
        assert (M.confusion_matrix(yc, pc) == sk.confusion_matrix(yc, pc)).all()
        assert M.classification_report(yc, pc, digits=5).split() == sk.classification_report(yc, pc, digits=5).split()


def test_score_prints_the_reference_lines():
    rs = np.random.RandomState(1)
    y = rs.uniform(-3, 3, size=50)
    p = y + rs.normal(0, 0.5, size=50)
    buf = io.StringIO()
    res = M.score(p, y, out=buf)
    text = buf.getvalue()
    for key in ("mae: ", "corr: ", "mult_acc: ", "mult f_score: ", "Confusion Matrix :", "Classification Report :", "Accuracy "):
        assert key in text
    assert abs(res["mae"] - np.mean(np.abs(p - y))) < 1e-12
    buf = io.StringIO()
    logits = rs.normal(size=(40, 3))
    yc = rs.randint(0, 3, size=40)
    r2 = M.score_classes(logits, yc, out=buf)
    assert abs(r2["accuracy"] - np.mean(np.argmax(logits, 1) == yc)) < 1e-12
    assert "Confusion Matrix :" in buf.getvalue() and "Accuracy " in buf.getvalue()
