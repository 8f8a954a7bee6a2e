"""Drop-in replacements for the native modules of the reference package
`template_speech_rec` (same module names, same `def` signatures, same dtypes,
return shapes and quirks), running on an MI355X through libtsr_hip.so:

    compute_likelihood_linear_filter   (compute_likelihood_linear_filter.pyx)
    code_parts                         (code_parts.pyx)
    spread_waliji_patches              (spread_waliji_patches.pyx)

plus the thin Python surface that sits directly on them in the reference:
`classification.Classifier`, `estimate_template.construct_linear_filter(s)`,
`test_template.score_template_background_section` and the per-utterance glue of
`get_train_data` (`_compute_detection_E`, `get_part_features`,
`get_waliji_feature_map`, `reorg_part_for_fast_filtering`, `collapse_to_grid`).

Put `template-speech-recognition_amd/` on sys.path and the reference's imports
(`import template_speech_rec.code_parts as cp`, ...) resolve here.
"""
