#!/usr/bin/env bash
# Same flags as the reference's GDR_model/infer.sh:10-15 (including --trivia, which its own parser rejects).
# INFER_CKPT: a Lightning .ckpt / state_dict; a path that does not exist is an error (as in the reference).  Leave it
# empty to run the seeded synthetic weights on the synthetic NQ-320k-shaped workload (no checkpoint ships with GDR).
INFER_CKPT=${INFER_CKPT:-}
BEAM_SIZE=${BEAM_SIZE:-100}
cd "$(dirname "$0")/.." || exit 1
python -m gdr_amd.main --decode_embedding 2 --n_gpu 1 --mode eval --query_type gtq_doc_aug_qg --adaptor_layer_num 4 \
--infer_ckpt "$INFER_CKPT" --num_return_sequences "$BEAM_SIZE" --tree 1 \
--model_info base --train_batch_size 64 --eval_batch_size 1 --test1000 0 --dropout_rate 0.1 --Rdrop 0.1 \
--adaptor_decode 1 --adaptor_efficient 1 --aug_query 1 --aug_query_type corrupted_query --input_dropout 1 --id_class bert_k30_c30_1 \
--kary 30 --output_vocab_size 30 --doc_length 64 --denoising 0 --max_output_length 10 \
--trivia 0 --nq 1 "$@"
