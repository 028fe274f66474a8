export GYP_TEST_HOOKS=1 GYP_SURVEY_SEED=0
O=gpurun_out/r06zo; mkdir -p $O; : > $O/surveys.txt
run() { echo "== $*" >> $O/surveys.txt; timeout 420 "$@" 2>&1 | grep -v "^$" | cut -c1-1600 | grep "^\[\|^    \|^{" | tail -12 >> $O/surveys.txt; tail -1 $O/surveys.txt | cut -c1-600; }
run python tools/bank_survey.py 6 8184000 41 12 1809 46100000
run python tools/bank_survey.py 6 2046000 26 10 2209 46200000
run python tools/big_survey.py 200 GYP_NO_SPEC 8184000 46700000 lock 6
