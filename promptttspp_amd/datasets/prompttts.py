"""Collator producing the reference's batch layout (promptttspp/datasets/prompttts.py:116-168):
(spks, utt_ids, phoneme (B,Tp) i64, duration (B,1,Tp) f32, phone_lengths (B) i64,
 mel (B,80,Tf) f32, log_cf0 (B,1,Tf), vuv (B,1,Tf), energy (B,1,Tf), frame_lengths (B) i64, prompts)
Prompts may be strings or pre-tokenised id lists; the latter are padded into an
(input_ids, attention_mask) pair (offline BERT vocabulary, see modules/prompt_encoder.py)."""
import torch


class PromptTTSCollator:
    def __call__(self, batch):
        spks, utt_ids, phonemes, durations, mels, log_cf0s, vuvs, energies, prompts = tuple(zip(*batch))
        B = len(spks)
        plen = [x.size(-1) for x in phonemes]
        flen = [x.size(-1) for x in mels]
        Tp, Tf, M = max(plen), max(flen), mels[0].size(0)
        phone = torch.zeros(B, Tp, dtype=torch.long)
        dur = torch.zeros(B, 1, Tp)
        mel = torch.zeros(B, M, Tf)
        cf0, vuv, energy = torch.zeros(B, 1, Tf), torch.zeros(B, 1, Tf), torch.zeros(B, 1, Tf)
        for i in range(B):
            p, f = plen[i], flen[i]
            phone[i, :p] = phonemes[i]
            dur[i, :, :p] = durations[i]
            mel[i, :, :f] = mels[i]
            cf0[i, :, :f], vuv[i, :, :f], energy[i, :, :f] = log_cf0s[i], vuvs[i], energies[i]
        if isinstance(prompts[0], torch.Tensor):  # pre-tokenised
            L = max(int(p.numel()) for p in prompts)
            ids = torch.zeros(B, L, dtype=torch.long)
            am = torch.zeros(B, L, dtype=torch.long)
            for i, p in enumerate(prompts):
                ids[i, : p.numel()] = p
                am[i, : p.numel()] = 1
            prompts = (ids, am)
        return (spks, utt_ids, phone, dur, torch.LongTensor(plen), mel, cf0, vuv, energy, torch.LongTensor(flen), prompts)
