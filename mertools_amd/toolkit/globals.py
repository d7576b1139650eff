"""Emotion <-> index tables (mirror of MERBench/toolkit/globals.py:1-5 — the label/index path must be bit-exact)."""
emos_mer = ['neutral', 'angry', 'happy', 'sad', 'worried', 'surprise']
emo2idx_mer = {emo: ii for ii, emo in enumerate(emos_mer)}
idx2emo_mer = {ii: emo for ii, emo in enumerate(emos_mer)}


# ---- MER2024 feature ranking for --model attention_topn (MER2024/toolkit/globals.py:218-231; low -> high, display names of the
# baseline paper) and the feature-directory stem behind each display name (the entries of featname_mapping, :145-203, that the
# rankings use).  Data of the benchmark, not code: Data_Feat_TOPN takes the last n of each list.
AUDIO_RANK_LOW2HIGH = ['eGeMAPS', 'VGGish', 'Whisper-base', 'emotion2vec', 'Whisper-large', 'wav2vec 2.0-base', 'wav2vec 2.0-large',
                       'HUBERT-base', 'HUBERT-large']
TEXT_RANK_LOW2HIGH = ['XLNet-base', 'ELECTRA-large', 'MOSS-7B', 'PERT-large', 'PERT-base', 'LERT-large', 'ELECTRA-base', 'LERT-base',
                      'RoBERTa-base', 'MacBERT-base', 'RoBERTa-large', 'ChatGLM2-6B', 'MacBERT-large', 'BLOOM-7B', 'Baichuan-13B']
IMAGE_RANK_LOW2HIGH = ['VideoMAE-base', 'EmoNet', 'VideoMAE-large', 'DINOv2-large', 'SENet-FER2013', 'ResNet-FER2013', 'MANet-RAFDB',
                       'EVA-02-base', 'CLIP-base', 'VideoMAE-base (VoxCeleb2)', 'VideoMAE-base (MER2023)', 'CLIP-large']
FEATURE_DIR_OF = {
    'eGeMAPS': 'eGeMAPS', 'VGGish': 'vggish', 'Whisper-base': 'whisper-base', 'emotion2vec': 'emotion2vec', 'Whisper-large': 'whisper-large-v2',
    'wav2vec 2.0-base': 'chinese-wav2vec2-base', 'wav2vec 2.0-large': 'chinese-wav2vec2-large', 'HUBERT-base': 'chinese-hubert-base',
    'HUBERT-large': 'chinese-hubert-large',
    'XLNet-base': 'chinese-xlnet-base', 'ELECTRA-large': 'chinese-electra-180g-large', 'MOSS-7B': 'moss-base-7b', 'PERT-large': 'chinese-pert-large',
    'PERT-base': 'chinese-pert-base', 'LERT-large': 'chinese-lert-large', 'ELECTRA-base': 'chinese-electra-180g-base', 'LERT-base': 'chinese-lert-base',
    'RoBERTa-base': 'chinese-roberta-wwm-ext', 'MacBERT-base': 'chinese-macbert-base', 'RoBERTa-large': 'chinese-roberta-wwm-ext-large',
    'ChatGLM2-6B': 'chatglm2-6b', 'MacBERT-large': 'chinese-macbert-large', 'BLOOM-7B': 'bloom-7b1', 'Baichuan-13B': 'Baichuan-13B-Base',
    'VideoMAE-base': 'videomae-base', 'EmoNet': 'emonet', 'VideoMAE-large': 'videomae-large', 'DINOv2-large': 'dinov2-large',
    'SENet-FER2013': 'senet50face', 'ResNet-FER2013': 'resnet50face', 'MANet-RAFDB': 'manet', 'EVA-02-base': 'eva02_base_patch14_224',
    'CLIP-base': 'clip-vit-base-patch32', 'VideoMAE-base (VoxCeleb2)': 'videomae-base-VoxCeleb2-99',
    'VideoMAE-base (MER2023)': 'videomae-base-K400-mer2023-299', 'CLIP-large': 'clip-vit-large-patch14',
}
