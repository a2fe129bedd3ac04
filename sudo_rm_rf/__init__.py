"""Import-path compatibility shim: ``sudo_rm_rf.dnn...`` resolves to the MI355X-native
implementation in ``sudo_rm_rf_amd`` so that the reference's README recipe (README.md:70-72) and
whole-module pickles (class path ``sudo_rm_rf.dnn.models.improved_sudormrf.SuDORMRF``) work
unchanged.  No code lives here."""
