"""Evaluation-side bookkeeping of the reference's hooks (speechbrain_convae_train.py:130-149,278-306):

  * ``AccuracyStats``            -- speechbrain.utils.Accuracy.AccuracyStats as the reference uses it
                                    (``append(log_probs.unsqueeze(0), labels.unsqueeze(0), length)``,
                                    ``summarize()`` = correct / total), counted on the device;
  * ``SimilarityMetricsStats``   -- utils/utility_similarity_aggregator.py:4-53 (running sum of the
                                    per-utterance encoder cosine similarities: "Utility_Retention").
"""
import torch


class AccuracyStats:
    def __init__(self):
        self.correct = 0
        self.total = 0

    def append(self, log_probabilities, targets, length=None):
        """log_probabilities [..., classes]; targets [...] (the reference passes both with a leading
        singleton batch axis and `length` = the number of utterances, which selects all of them)."""
        pred = log_probabilities.argmax(dim=-1).reshape(-1)
        tgt = torch.as_tensor(targets, device=pred.device).reshape(-1)
        self.correct = self.correct + (pred == tgt).sum()
        self.total += int(tgt.numel())

    def summarize(self):
        return float(self.correct) / max(1, self.total)


class SimilarityMetricsStats:
    def __init__(self):
        self.clear()

    def clear(self):
        self.scores, self.summary = [], {}
        self.value, self.denom = 0, 0

    def append(self, scores):
        scores = scores.detach()
        self.scores.extend(scores)
        self.value = self.value + torch.sum(scores)
        self.denom += scores.shape[0]

    def peek(self):
        return self.value / (1.0 * self.denom)

    def summarize(self):
        if isinstance(self.scores, list):
            self.scores = torch.stack(self.scores)
        self.summary["average"] = torch.sum(self.scores) / self.scores.shape[0]
        return self.summary["average"]
